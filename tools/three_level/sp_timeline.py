"""Round 5: where a three-level solve spends its wall time, from a `rocprofv3 --kernel-trace --output-format csv` run (KEEP_TRACE=1
tools/jobs/kernel_stats.sh ...): the wall clock of the LAST solve in the trace partitioned by what the chip was doing, in priority order
product (k_mul16k / k_xor16 / k_zero16) > replay and outer passes (k_update16k, k_outer_apply) > gather / clear (k_gather_b,
k_zero_dead_mults) > one-level bulk (k_update16, k_block_trsm) > panel path (everything else) > idle, and the same per super-panel
(a super-panel ends with its last k_mul16k / k_xor16).  usage: sp_timeline.py <trace dir>"""
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
ev = []
for path in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    with open(path) as f:
        for r in csv.DictReader(f):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]))
ev.sort()
# the last solve: from the last k_to_tiled / k_synth on
starts = [i for i, e in enumerate(ev) if e[2] in ("k_to_tiled",)]
ev = ev[starts[-1]:] if starts else ev
def cls(n):
    if n in ("k_mul16k", "k_xor16", "k_zero16"): return 0
    if n in ("k_update16k", "k_outer_apply"): return 1
    if n in ("k_gather_b", "k_zero_dead_mults"): return 2
    if n in ("k_update16", "k_block_trsm"): return 3
    return 4
names = ["product", "outer pass / replay", "gather + clear", "one-level bulk", "panel path + rest", "idle"]
pts = []
for s, e, n in ev:
    pts.append((s, 0, cls(n))); pts.append((e, 1, cls(n)))
pts.sort()
active = [0] * 5
tot = defaultdict(float)
t_prev = pts[0][0]
segs = []          # (t0, t1, class)
for t, kind, c in pts:
    if t > t_prev:
        cur = next((k for k in range(5) if active[k] > 0), 5)
        tot[cur] += t - t_prev
        segs.append((t_prev, t, cur))
    active[c] += 1 if kind == 0 else -1
    t_prev = t
wall = (pts[-1][0] - pts[0][0]) / 1e6
print(f"# last solve in {root}: {wall:.1f} ms of kernels' wall time")
for k in range(6):
    print(f"  {names[k]:24s} {tot[k] / 1e6:9.2f} ms  {100 * tot[k] / 1e6 / wall:5.1f} %")
per = defaultdict(float)
for s, e, n in ev:
    per[n] += (e - s) / 1e6
print("# kernel time by name (ms, summed over streams): " + ", ".join(f"{n} {v:.1f}" for n, v in sorted(per.items(), key=lambda x: -x[1])[:14]))
# per super-panel: split at the end of each run of product kernels (gap to the next product kernel > 5 ms)
prod = [(s, e) for s, e, n in ev if cls(n) == 0]
if prod:
    ends = []
    for i, (s, e) in enumerate(prod):
        if i + 1 == len(prod) or prod[i + 1][0] - e > 5e6: ends.append(e)
    t0 = pts[0][0]
    print("# per super-panel (ms): product | outer/replay | gather | one-level bulk | panel path | idle")
    for k, te in enumerate(ends + [pts[-1][0]]):
        acc = [0.0] * 6
        for a, b, c in segs:
            lo, hi = max(a, t0), min(b, te)
            if hi > lo: acc[c] += (hi - lo) / 1e6
        print(f"  {'tail' if k == len(ends) else 'super-panel ' + str(k):14s} {(te - t0) / 1e6:8.1f} ms: " + " | ".join(f"{v:7.1f}" for v in acc))
        t0 = te
