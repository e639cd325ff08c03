"""Per-launch HBM traffic of the sparse-TSDF kernels from the rocprofv3 --pmc passes of tools/prof_tsdf_r06.sh, beside the kernel-stats durations and the
algorithmic bytes of the two benchmarks (40 B per UPDATED voxel + 16 B per pixel).  traffic = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE
counts 128-B requests as 64 B on gfx950 (MI355X_MICROARCH.md).  usage: python tools/tsdf_pmc.py <dir with pmc_<bench>_<COUNTER>/ and <bench>_kernel_stats.csv>"""
import collections, csv, glob, json, os, sys

d = sys.argv[1]
out = {}
for bench in ("tsdf_sparse", "tile_tail", "tile_tail_terrain"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob(os.path.join(d, f"pmc_{bench}_{c}", "**", "*_counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0].replace("void ", "")
                if k.startswith("k_ts_"):
                    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = {}
    ks = os.path.join(d, f"{bench}_kernel_stats.csv")
    if os.path.exists(ks):
        for r in csv.DictReader(open(ks)):
            dur[r["Name"].split("(")[0].replace("void ", "")] = (float(r["AverageNs"]) / 1e3, int(r["Calls"]))
    res = {}
    for k, v in sorted(agg.items()):
        m = {c: sum(x) / len(x) for c, x in v.items()}
        e = {"FETCH_SIZE_KiB": round(m.get("FETCH_SIZE", 0.0), 1), "WRITE_SIZE_KiB": round(m.get("WRITE_SIZE", 0.0), 1),
             "hbm_bytes_per_launch": int((2.0 * m.get("FETCH_SIZE", 0.0) + m.get("WRITE_SIZE", 0.0)) * 1024.0),
             "lines_fetched_128B": int(m.get("FETCH_SIZE", 0.0) * 1024.0 / 64.0)}      # FETCH_SIZE tallies every fetched 128-byte line as 64 B (profiles/r06_hbm_granule.json)
        if k in dur:
            e["avg_us"] = round(dur[k][0], 2); e["calls"] = dur[k][1]
            e["traffic_GBps"] = round(e["hbm_bytes_per_launch"] / (dur[k][0] * 1e-6) / 1e9, 1)
            if e["lines_fetched_128B"]:
                e["ps_per_fetched_line"] = round(dur[k][0] * 1e6 / e["lines_fetched_128B"], 1)
        res[k] = e
    bj = os.path.join(d, f"{bench}.json")
    if os.path.exists(bj):
        try:
            b = json.loads(open(bj).read().strip().splitlines()[-1])
            alg = b.get("algorithmic_bytes_per_frame") or (b.get("algorithmic_GB_per_frame") or 0) * 1e9
            if alg and "k_ts_integrate_col" in res and "avg_us" in res["k_ts_integrate_col"]:
                r = res["k_ts_integrate_col"]
                r["algorithmic_bytes_per_frame"] = int(alg)
                r["algorithmic_GBps"] = round(alg / (r["avg_us"] * 1e-6) / 1e9, 1)
                r["frac_of_8TBps"] = round(alg / (r["avg_us"] * 1e-6) / 8e12, 3)
                r["traffic_over_algorithmic"] = round(r["hbm_bytes_per_launch"] / alg, 2)
        except Exception as ex:      # noqa: BLE001
            res["_error"] = str(ex)
    out[bench] = res
print(json.dumps(out, indent=1))
