"""Per-iteration kernel times from a rocprofv3 --kernel-trace --stats csv:  python tools/kstats.py <kernel_stats.csv> [iterations=110] [rows=26]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
it = float(sys.argv[2]) if len(sys.argv) > 2 else 110.0
for r in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 26]:
    print("%-80s %5d %8.2fus %8.2fus/iter" % (r["Name"][:80], int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / it / 1e3))
