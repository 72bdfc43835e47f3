"""Per-outer-panel timeline of ONE two-level solve from a rocprofv3 --kernel-trace --output-format csv run:
where the ~0.4-0.8 ms per panel outside the inner elimination and the outer pass go.
usage: outer_timeline.py <dir> [every]"""
import csv, glob, os, sys
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True))[0]
every = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
S = lambda r: int(r["Start_Timestamp"]); E = lambda r: int(r["End_Timestamp"])
nm = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "")
# the LAST solve in the trace (profile_one: warm-up solves first): from the last k_to_tiled on
starts = [i for i, r in enumerate(rows) if nm(r).startswith("k_to_tiled")]
rows = rows[starts[-1]:]
t0 = S(rows[0])
prow = [r for r in rows if nm(r) == "k_outer_prow"]
print(f"solve: {len(rows)} launches, {len(prow)} outer panels, span {(max(E(r) for r in rows) - t0) / 1e6:.2f} ms")
print("panel  t_prow   inner_ms(since prev prow)  fast#  prow+trsm_us  apply_us(n)  k16k_us(n)  k16k_end-t_prow  update16_us(n)  idle_us")
prev = t0
for p, pr in enumerate(prow):
    nxt = S(prow[p + 1]) if p + 1 < len(prow) else max(E(r) for r in rows)
    seg = [r for r in rows if prev <= S(r) < S(pr)]                # the panel's inner elimination
    aft = [r for r in rows if S(pr) <= S(r) < nxt]
    fast = [r for r in seg if nm(r) == "k_block_fast"]
    trsm = [r for r in aft if nm(r).startswith("k_outer_trsm")]
    app = [r for r in aft if nm(r) == "k_outer_apply"]
    k16k = [r for r in aft if nm(r).startswith("k_update16k")]
    u16 = [r for r in seg if nm(r).startswith("k_update16<")]
    if p % every == 0 or p < 3:
        dur = lambda L: sum(E(r) - S(r) for r in L) / 1e3
        print(f"{p:5d} {(S(pr) - t0) / 1e6:7.2f} {(S(pr) - prev) / 1e6:10.3f} {len(fast):12d} {((E(trsm[-1]) if trsm else E(pr)) - S(pr)) / 1e3:10.1f} "
              f"{dur(app):9.1f}({len(app)}) {dur(k16k):9.1f}({len(k16k)}) {((max(E(r) for r in k16k) if k16k else S(pr)) - S(pr)) / 1e3:12.1f} {dur(u16):12.1f}({len(u16)})")
    prev = S(pr)
for name in ("k_outer_prow", "k_outer_trsm", "k_outer_apply", "k_update16k", "k_update16<", "k_block_fast", "k_narrow_all", "k_prio_window", "k_block_trsm", "k_gate"):
    L = [r for r in rows if nm(r).startswith(name)]
    if L: print(f"{name:16s} {len(L):6d} launches  {sum(E(r) - S(r) for r in L) / 1e6:8.2f} ms total  {sum(E(r) - S(r) for r in L) / len(L) / 1e3:8.1f} us avg")
