"""Per-system time budget of the batch workload from a `rocprofv3 --kernel-trace --output-format csv` run of
tools/profile_batch.py (BASELINE configs[3]).  Wall time of the chosen repetition is PARTITIONED by what the chip was doing,
in this order of precedence: bulk update running (k_update16 / k_update16k) > outer-panel preparation (k_outer_*) > TRSM >
panel path (search, narrow step, look-ahead) > back-substitution > pack / export copies > gates only > idle.
usage: gang_budget.py <trace dir> <nsys per repetition> [repetition index, default last]"""
import csv, glob, os, sys
root, nsys = sys.argv[1], int(sys.argv[2])
rep = int(sys.argv[3]) if len(sys.argv) > 3 else -1
rows = []
for path in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""),
                         r["Queue_Id"]))
rows.sort()
CLASSES = [("bulk update", ("k_update16",)), ("outer-panel prep", ("k_outer_",)), ("TRSM", ("k_block_trsm",)),
           ("panel path", ("k_block_fast", "k_panel_step", "k_narrow_all", "k_prio_window", "k_win_", "k_unwind", "k_check_rhs")),
           ("back-substitution", ("k_bs_",)), ("pack / export", ("k_to_tiled", "k_pack", "__amd_rocclr", "k_synth")),
           ("gates only", ("k_gate", "k_probe"))]
def cls(name):
    for i, (_, pats) in enumerate(CLASSES):
        if any(name.startswith(p) for p in pats):
            return i
    return len(CLASSES) - 1
# repetitions: a k_to_tiled more than 5 ms after the previous kernel's END starts... no: split at the largest idle gaps before a k_to_tiled
starts = [s for s, e, n, q in rows if n.startswith("k_to_tiled")]
reps, last_end = [], 0
cuts = []
for s, e, n, q in rows:
    if n.startswith("k_synth"):
        continue
    if n.startswith("k_to_tiled") and (not cuts or s - last_end > 3_000_000) and not any(abs(s - c) < 20_000_000 for c in cuts):
        cuts.append(s)
    last_end = max(last_end, e)
cuts.append(last_end + 1)
lo, hi = cuts[rep if rep >= 0 else len(cuts) - 2], cuts[(rep if rep >= 0 else len(cuts) - 2) + 1]
win = [(s, e, n, q) for s, e, n, q in rows if s >= lo - 2_000_000 and s < hi and not n.startswith("k_synth")]
lo = min(s for s, e, n, q in win); hi = max(e for s, e, n, q in win)
ev = []
for s, e, n, q in win:
    c = cls(n)
    ev.append((s, 1, c)); ev.append((e, -1, c))
ev.sort()
active = [0] * len(CLASSES)
part = [0.0] * (len(CLASSES) + 1)
prev = lo
for t, d, c in ev:
    top = next((i for i in range(len(CLASSES)) if active[i] > 0), len(CLASSES))
    part[top] += t - prev
    prev = t
    active[c] += d
wall = hi - lo
print(f"# {root}: repetition window {wall / 1e6:.1f} ms, {nsys} systems -> {wall / 1e6 / nsys:.3f} ms per system = {nsys / (wall / 1e9):.0f} systems/s")
print(f"# {'what the chip was doing':28s} {'ms':>9} {'share':>7} {'ms/system':>10}")
for i, (name, _) in enumerate(CLASSES + [("idle (host, launch gaps)", ())]):
    print(f"  {name:28s} {part[i] / 1e6:9.2f} {100 * part[i] / wall:6.1f}% {part[i] / 1e6 / nsys:10.3f}")
busy = {}
for s, e, n, q in win:
    busy.setdefault(n[:30], [0, 0.0]); busy[n[:30]][0] += 1; busy[n[:30]][1] += e - s
print("# kernel time summed over launches (launches of different gangs / streams overlap):")
for n, (c, t) in sorted(busy.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"  {n:32s} {c:6d} launches {t / 1e6:9.2f} ms  avg {t / c / 1e3:8.1f} us")
