"""Bytes per 16-byte access that rocprofv3's FETCH_SIZE / WRITE_SIZE report for the access patterns of tools/microbench/hbm_granule.hip.
usage: python tools/hbm_granule_report.py <dir holding pmc_FETCH_SIZE/ and pmc_WRITE_SIZE/ and optionally stats/>  -> JSON"""
import collections, csv, glob, json, os, re, sys

d = sys.argv[1]
N = 16 << 20
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(os.path.join(d, f"pmc_{c}", "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"(rd|wrw|rmww|wr|rmw)<([\d, ]+)>", r["Kernel_Name"])
            if m:
                agg[f"{m.group(1)}<{m.group(2).replace(' ', '')}>"][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = {}
for f in glob.glob(os.path.join(d, "stats", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(rd|wrw|rmww|wr|rmw)<([\d, ]+)>", r["Name"])
        if m:
            dur[f"{m.group(1)}<{m.group(2).replace(' ', '')}>"] = float(r["AverageNs"]) / 1e3
out = {}
for k in sorted(agg, key=lambda s: (s.split("<")[0], [int(x) for x in s.split("<")[1][:-1].split(",")])):
    v = {c: sum(x) / len(x) for c, x in agg[k].items()}
    e = {"FETCH_SIZE_bytes_per_access": round(v.get("FETCH_SIZE", 0.0) * 1024 / N, 2), "WRITE_SIZE_bytes_per_access": round(v.get("WRITE_SIZE", 0.0) * 1024 / N, 2)}
    if k in dur:
        e["avg_us"] = round(dur[k], 1)
        width = int(k.split("<")[1].split(",")[0]) if k.startswith(("wrw", "rmww")) else 16
        e["useful_GBps"] = round(N * width * (2 if k.startswith("rmw") else 1) / dur[k] / 1e3, 1)
        stride = int(k[:-1].split("<")[1].split(",")[-1])
        e["ps_per_128B_line"] = round(dur[k] * 1e6 / (N * stride / 128.0), 1) if stride >= 128 else round(dur[k] * 1e6 / (N * stride / 128.0), 1)
    out[k] = e
print(json.dumps({"what": "rocprofv3 counter bytes per 16-byte access, one access per <stride> bytes, 16 Mi accesses over 4 GiB (MI355X)", "kernels": out}, indent=1))
