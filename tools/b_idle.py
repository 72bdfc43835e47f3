"""Per-block view of the bulk stream from a rocprofv3 --kernel-trace --output-format csv run of ONE solve:
duration of every 16th block's TRSM and update, and how long the bulk stream sat idle before the TRSM.
usage: b_idle.py <dir>"""
import csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
S = lambda r: int(r["Start_Timestamp"]); E = lambda r: int(r["End_Timestamp"])
upd = sorted((r for r in rows if "k_update" in r["Kernel_Name"]), key=S)
trs = sorted((r for r in rows if "k_block_trsm" in r["Kernel_Name"]), key=S)
n = min(len(upd), len(trs))
print("block  update_us  trsm_us  idle_before_trsm_us")
for b in range(0, n, max(1, n // 16)):
    idle = (S(trs[b]) - E(upd[b - 1])) / 1e3 if b else 0.0
    print(f"{b:5d} {(E(upd[b]) - S(upd[b])) / 1e3:10.1f} {(E(trs[b]) - S(trs[b])) / 1e3:8.1f} {idle:10.1f}")
tot_idle = sum((S(trs[b]) - E(upd[b - 1])) / 1e3 for b in range(1, n))
print(f"bulk stream: update {sum(E(r) - S(r) for r in upd) / 1e6:.2f} ms, trsm {sum(E(r) - S(r) for r in trs) / 1e6:.2f} ms, "
      f"idle before trsm {tot_idle / 1e3:.2f} ms, span {(E(upd[n - 1]) - S(trs[0])) / 1e6:.2f} ms")
