"""Sum rocprofv3 --pmc counters per kernel from a --output-format csv run.
usage: pmc_summary.py <dir> [kernel-substring]   (reads every *counter_collection.csv below <dir>)"""
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(float))
calls = defaultdict(set)
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row.get("Kernel_Name", "")
            if want not in name:
                continue
            short = name.split("(")[0][:60]
            acc[short][row["Counter_Name"]] += float(row["Counter_Value"])
            calls[short].add(row.get("Dispatch_Id"))
for k, d in acc.items():
    print(k, "dispatches", len(calls[k]))
    for c in sorted(d):
        print(f"   {c:28s} {d[c]:.4e}")
