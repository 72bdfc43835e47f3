"""Per-kernel table (calls, total, average, share) from a `rocprofv3 --kernel-trace --stats --output-format csv` run:
the *kernel_stats.csv when rocprofv3 wrote one, else summed from *kernel_trace.csv.  usage: rocprof_csv_summary.py <dir> [title]"""
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
title = sys.argv[2] if len(sys.argv) > 2 else root
rows = []
for path in glob.glob(os.path.join(root, "**", "*kernel_stats.csv"), recursive=True):
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((r["Name"].split("(")[0], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3,
                         float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
if not rows:
    acc = defaultdict(list)
    for path in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                acc[r["Kernel_Name"].split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    rows = [(k, len(v), sum(v), sum(v) / len(v), min(v), max(v)) for k, v in acc.items()]
rows.sort(key=lambda r: -r[2])
tot = sum(r[2] for r in rows) or 1.0
print(f"# rocprofv3 --kernel-trace --stats -- {title}")
print(f"# {'calls':>7} {'total_us':>12} {'avg_us':>10} {'min_us':>9} {'max_us':>10} {'pct':>6}  kernel")
for name, calls, total, avg, mn, mx in rows:
    print(f"  {calls:>7} {total:>12.1f} {avg:>10.2f} {mn:>9.2f} {mx:>10.2f} {100 * total / tot:>6.2f}  {name[:90]}")
