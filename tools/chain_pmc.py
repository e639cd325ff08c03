"""Per-launch HBM traffic of the forward's ordering chain at P = 3 M (tools/gpu_profile_r06.sh step 7): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
(<dir>/pmc3m_<COUNTER>/) beside the kernel-stats durations.  traffic = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (FETCH_SIZE tallies a fetched 128-byte line as 64 B,
WRITE_SIZE counts 32-byte sectors: profiles/r06_hbm_granule.json).  usage: python tools/chain_pmc.py <dir> <kernel_stats.csv>"""
import collections, csv, glob, json, os, sys

d, ks = sys.argv[1], sys.argv[2]
short = lambda n: n.split("(")[0].replace("void ", "").strip()
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(os.path.join(d, f"pmc3m_{c}", "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = {}
if os.path.exists(ks):
    for r in csv.DictReader(open(ks)):
        dur[short(r["Name"])] = (float(r["AverageNs"]) / 1e3, int(r["Calls"]))
out = {}
for k, v in sorted(agg.items()):
    if not k.startswith("k_") or k.startswith("k_debug"):
        continue
    m = {c: sum(x) / len(x) for c, x in v.items()}
    e = {"FETCH_SIZE_KiB": round(m.get("FETCH_SIZE", 0.0), 1), "WRITE_SIZE_KiB": round(m.get("WRITE_SIZE", 0.0), 1),
         "hbm_bytes_per_launch": int((2.0 * m.get("FETCH_SIZE", 0.0) + m.get("WRITE_SIZE", 0.0)) * 1024.0)}
    if k in dur:
        e["avg_us"] = round(dur[k][0], 2); e["calls"] = dur[k][1]
        e["traffic_GBps"] = round(e["hbm_bytes_per_launch"] / (dur[k][0] * 1e-6) / 1e9, 1)
        e["frac_of_8TBps"] = round(e["traffic_GBps"] / 8000.0, 3)
    out[k] = e
print(json.dumps({"what": "bench.py --variant surfel --P 3000000, 1920x1080: every k_* kernel of one training iteration, per launch", "kernels": out}, indent=1))
