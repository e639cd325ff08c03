"""Aggregates gpurun_out/prof_bwd/*: per-launch averages of the SQ counters for the blend kernels."""
import collections, csv, glob, json, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(root, "gpurun_out", "prof_bwd")
res = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sorted(glob.glob(os.path.join(O, "*_[0-9]"))):
    mode = os.path.basename(d).split("_")[0]
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "k_blend" not in k:
                continue
            name = k.split("(")[0].replace("void ", "")
            res[mode + ":" + name][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: round(sum(v) / len(v), 1) for c, v in sorted(cs.items())} for k, cs in sorted(res.items())}
json.dump(out, open(os.path.join(O, "summary.json"), "w"), indent=1)
for k, cs in out.items():
    print(k)
    for c, v in cs.items():
        print(f"   {c:28s} {v:16.1f}")
