"""Builds profiles/traffic.json + a per-kernel PMC summary from rocprofv3 --pmc passes (run tools/gpu_profile.sh first).

HBM traffic per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 bytes: FETCH_SIZE/WRITE_SIZE are in KiB and, on gfx950 with
this rocprofv3, FETCH_SIZE counts 128-B read requests as 64 B (MI355X_MICROARCH.md, "HBM"): the read side is doubled
as that section prescribes.  WRITE_SIZE is uncalibrated there and used as is.  Infinity-Cache hits are included."""
import collections, csv, glob, json, os, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "gpurun_out")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
files = []
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
for d in glob.glob(os.path.join(out, "pmc_%s_*" % tag if tag != "r01" else "pmc_*", "*")):
    cands = sorted(glob.glob(os.path.join(d, "*_counter_collection.csv")), key=os.path.getmtime)
    if cands:
        files.append(cands[-1])          # newest pass only (gpurun_out accumulates across calls)
for f in files:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if not (k.startswith("void k_") or k.startswith("k_")):
            continue
        name = k.split("(")[0].replace("void ", "").replace(", false>", ">").replace(", true>", ",mfma>")
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = {"k_blend_fwd<0>": "ewa:blend_fwd", "k_blend_fwd<1>": "surfel:blend_fwd", "k_blend_fwd<2>": "plane:blend_fwd",
         "k_blend_bwd<0>": "ewa:blend_bwd", "k_blend_bwd<1>": "surfel:blend_bwd", "k_blend_bwd<2>": "plane:blend_bwd",
         "k_blend_bwd_sp<0>": "ewa:blend_bwd_sp", "k_blend_bwd_sp<1>": "surfel:blend_bwd_sp", "k_blend_bwd_sp<2>": "plane:blend_bwd_sp"}
traffic = {}
tj = os.path.join(root, "profiles", "traffic.json")
if os.path.exists(tj):
    traffic = json.load(open(tj))
summary = {}
for k, v in sorted(agg.items()):
    m = {c: sum(x) / len(x) for c, x in v.items()}
    summary[k] = {c: round(val, 1) for c, val in sorted(m.items())}
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        hbm = (2.0 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024.0
        summary[k]["hbm_bytes_per_launch"] = int(hbm)
        if k in names:
            traffic[names[k]] = int(hbm)
json.dump(traffic, open(tj, "w"), indent=1, sort_keys=True)
json.dump(summary, open(os.path.join(root, "profiles", f"{tag}_pmc_summary.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(traffic, indent=1))
