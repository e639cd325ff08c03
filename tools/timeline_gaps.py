"""Reads a rocprofv3 --kernel-trace CSV (kernel_trace.csv) and prints, for the steady-state part of the run, what one iteration is made of:
kernels in launch order with duration and the idle gap in front of each, the sum of durations, the sum of gaps and the period.
usage: python tools/timeline_gaps.py <dir with *_kernel_trace.csv> [anchor kernel substring = k_preprocess_surfel]"""
import csv
import glob
import json
import sys


def main():
    d = sys.argv[1]
    anchor = sys.argv[2] if len(sys.argv) > 2 else "k_preprocess_surfel"
    f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
    rows = list(csv.DictReader(open(f)))
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda e: e[0])
    starts = [i for i, e in enumerate(ev) if anchor in e[2]]
    if len(starts) < 12:
        print("too few iterations", len(starts)); return
    lo, hi = starts[len(starts) // 2], starts[len(starts) // 2 + 8]          # 8 iterations from the middle of the run
    seg = ev[lo:hi]
    period = (ev[hi][0] - ev[lo][0]) / 8
    per = {}
    order = []
    prev_end = None
    for s, e, n in seg:
        n = n.split("(")[0][:60]
        if n not in per:
            per[n] = [0, 0, 0]; order.append(n)
        per[n][0] += 1; per[n][1] += e - s
        if prev_end is not None:
            per[n][2] += max(0, s - prev_end)
        prev_end = max(prev_end or 0, e)
    tot_d = sum(v[1] for v in per.values()) / 8; tot_g = sum(v[2] for v in per.values()) / 8
    out = {"period_us": round(period / 1e3, 1), "kernel_us": round(tot_d / 1e3, 1), "gap_us": round(tot_g / 1e3, 1), "launches_per_iter": len(seg) / 8,
           "kernels": [{"name": n, "per_iter": per[n][0] / 8, "dur_us": round(per[n][1] / per[n][0] / 1e3, 2), "gap_before_us": round(per[n][2] / per[n][0] / 1e3, 2)} for n in order]}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
