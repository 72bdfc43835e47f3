"""Per-pass rate of the bulk update inside ONE solve, from a rocprofv3 --kernel-trace --output-format csv run of
tools/profile_one.py N 1: duration and grid of every k_update16 launch of the last solve against the size of its pass
(dense N x N: rows alive x tiles right of the block), to compare with the isolated kernel (profiles/r03_slab_pad.txt,
tools/microbench_update16.hip: ~9 us fixed + 4.2-4.9 TB/s from HBM, 5.3-6.4 from the Infinity Cache).
usage: pass_rates.py <dir> N [every]"""
import csv, glob, gzip, os, sys
d, n = sys.argv[1], int(sys.argv[2]); every = int(sys.argv[3]) if len(sys.argv) > 3 else 16
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv*"), recursive=True))[0]
rows = list(csv.DictReader(gzip.open(f, "rt") if f.endswith(".gz") else open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
nm = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "")
last = [i for i, r in enumerate(rows) if nm(r).startswith("k_to_tiled")][-1]
rows = rows[last:]
upd = [r for r in rows if nm(r).startswith("k_update16<")]
S = lambda r: int(r["Start_Timestamp"]); E = lambda r: int(r["End_Timestamp"])
W = (n + 1 + 63) // 64
print(f"{len(upd)} bulk launches; block  wgs  MiB  us  TB/s(sweep-words)  gap_before_us(previous update end -> this start)")
tot_b = tot_t = 0.0
for b, r in enumerate(upd):
    rows_alive = n - 256 * b; words = W - 4 * (b + 1) - (4 if b + 1 < len(upd) else 0)      # the next window's words are left out
    by = max(rows_alive, 0) * max(words, 0) * 16.0
    us = (E(r) - S(r)) / 1e3
    tot_b += by; tot_t += us
    if b % every == 0 or b == len(upd) - 1:
        gap = (S(r) - E(upd[b - 1])) / 1e3 if b else 0.0
        print(f"{b:5d} {int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']):5d} {by / 2 / 2**20:7.1f} {us:8.1f} {by / us / 1e6:6.2f} {gap:8.1f}")
print(f"sum: {tot_b / 1e9:.1f} GB in {tot_t / 1e3:.2f} ms of kernel time = {tot_b / tot_t / 1e6:.2f} TB/s")
