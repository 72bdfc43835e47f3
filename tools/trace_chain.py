"""Critical-path view of a rocprofv3 --kernel-trace --output-format csv run: per kernel (and per grid size for
k_panel_step) count / mean duration, and the mean idle gap before it on its queue.
usage: trace_chain.py <dir>"""
import csv, glob, os, sys
from collections import defaultdict
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
byq = defaultdict(list)
for r in rows:
    byq[r["Queue_Id"]].append(r)
stat = defaultdict(lambda: [0, 0.0, 0.0])
for q, lst in byq.items():
    prev_end = None
    for r in lst:
        name = r["Kernel_Name"].split("(")[0]
        if "k_panel_step" in name:
            name += f" grid={r.get('Grid_Size_X')}"
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        st = stat[(q, name)]
        st[0] += 1; st[1] += (e - s) / 1e3
        if prev_end is not None:
            st[2] += max(0, s - prev_end) / 1e3
        prev_end = e
t0 = int(rows[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in rows)
print(f"span {(t1 - t0) / 1e6:.2f} ms, {len(rows)} kernels")
for (q, name), (n, dur, gap) in sorted(stat.items(), key=lambda kv: -kv[1][1]):
    print(f"queue {q:>3s} {name[:60]:60s} n={n:5d} mean {dur / n:8.2f} us  gap-before {gap / n:7.2f} us  total {dur / 1e3:8.2f} ms")
