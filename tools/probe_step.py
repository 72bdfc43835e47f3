"""Where does a panel step spend its time?  Builds the library with -DGF2_STEP_PROBE (wall-clock stamps inside
k_panel_step, see gf2_kernels.hip.h) next to this file, runs synthetic solves and prints, for the G+1 steps of the
chosen blocks, the phase times of the workgroups and of the search units.

    python tools/probe_step.py [N] [block ...]
"""
import ctypes, os, subprocess, sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "_probe", "libgf2bv_hip_probe.so")
SRC = os.path.join(ROOT, "gf2bv_amd", "csrc", "gf2_solver.hip")
if not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(SRC), os.path.getmtime(SRC.replace("gf2_solver.hip", "gf2_kernels.hip.h"))):
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
                           "-DGF2_STEP_PROBE", SRC, "-o", LIB])
if "--build-only" in sys.argv:
    sys.exit(0)
os.environ["GF2BV_LIB"] = LIB
sys.path.insert(0, ROOT)
import numpy as np
from gf2bv_amd import hip

args = [a for a in sys.argv[1:] if not a.startswith("-")]
n = int(args[0]) if args else 65536
blocks = [int(a) for a in args[1:]] or [8, n // 256 // 2, n // 256 - 8]
G, WGS, UNITS = 4, 384, 256
lib = hip.lib()
stride = hip.padded_stride(n)
buf = hip.DeviceBuffer(n * stride * 8)
hip.synth_device(buf.ptr, n, n, stride, 1234)
hip.solve_device(buf.ptr, n, n, stride, 0)          # warm-up
for b in blocks:
    hip.synth_device(buf.ptr, n, n, stride, 1234)
    assert lib.gf2bv_probe_set(ctypes.c_int(b * G)) == 0
    sol = hip.solve_device(buf.ptr, n, n, stride, 0)
    wg = np.zeros((G + 1, WGS, 4), dtype=np.uint64)
    un = np.zeros((G + 1, UNITS, 6), dtype=np.uint64)
    assert lib.gf2bv_probe_read(wg.ctypes.data_as(ctypes.c_void_p), un.ctypes.data_as(ctypes.c_void_p)) == 0
    wg = wg.astype(np.int64); un = un.astype(np.int64)
    upd = np.zeros((WGS, 6), dtype=np.uint64)
    assert lib.gf2bv_probe_read_update(upd.ctypes.data_as(ctypes.c_void_p)) == 0
    upd = upd.astype(np.int64)
    find_wgs = min(256, (n + 255) // 256 + 3) // 4 if n >= 256 else 1
    units = min(256, max(1, (n + 255) // 256)); find_wgs = (units + 3) // 4
    print(f"N={n} block {b} (rank {sol.rank}, eliminate {sol.stats['ms_eliminate']:.1f} ms); times in us from the step's first stamp")
    tprev_end = None
    for s in range(G + 1):
        w = wg[s]
        on = w[:, 0] > 0
        if not on.any():
            continue
        t0 = w[on, 0].min()
        us = lambda x: (x - t0) / 100.0
        line = f" step {s}: "
        if tprev_end is not None:
            line += f"gap from previous step's last stamp {us(tprev_end) * -1:.1f}; "
        fw = np.arange(WGS) < (find_wgs if s < G else 0)
        nar = on & ~fw
        fin = on & fw
        last = t0
        if fin.any():
            line += (f"search WGs {fin.sum()}: entry {us(w[fin, 0].min()):.1f}..{us(w[fin, 0].max()):.1f}, params in "
                     f"{us(w[fin, 1].min()):.1f}..{us(w[fin, 1].max()):.1f}, P built {us(w[fin, 2].min()):.1f}..{us(w[fin, 2].max()):.1f}; ")
            last = max(last, w[fin, 2].max())
        if nar.any():
            line += (f"narrow WGs {nar.sum()}: entry {us(w[nar, 0].min()):.1f}..{us(w[nar, 0].max()):.1f}, params in "
                     f"{us(w[nar, 1].min()):.1f}..{us(w[nar, 1].max()):.1f}, P built {us(w[nar, 2][w[nar, 2] > 0].min()) if (w[nar, 2] > 0).any() else -1:.1f}..{us(w[nar, 2].max()):.1f}, "
                     f"end {us(w[nar, 3][w[nar, 3] > 0].min()) if (w[nar, 3] > 0).any() else -1:.1f}..{us(w[nar, 3].max()):.1f}; ")
            last = max(last, w[nar, 3].max())
        print(line)
        if os.environ.get("PROBE_ENTRIES"):
            ent = np.sort(us(w[on, 0]))
            print("         entry times (every 16th workgroup, sorted):", " ".join(f"{x:.0f}" for x in ent[::16]))
        if s < G:
            u = un[s]
            act = u[:, 5] > 0
            pub = u[:, 4] > 0
            line = "         units: "
            if act.any():
                line += (f"{act.sum()} scanned (chunks {u[act, 5].min()}..{u[act, 5].max()}), loop start {us(u[act, 0].min()):.1f}..{us(u[act, 0].max()):.1f}, "
                         f"loop end {us(u[act, 1].min()):.1f}..{us(u[act, 1].max()):.1f}; ")
            arr = u[:, 2] > 0
            if arr.any():
                line += f"arrivals {us(u[arr, 2].min()):.1f}..{us(u[arr, 2].max()):.1f}; "
            if pub.any():
                k = int(np.argmax(u[:, 4]))
                line += f"publisher unit {k}: arrived {us(u[k, 2]):.1f}, decided {us(u[k, 3]):.1f}, published {us(u[k, 4]):.1f}"
                last = max(last, u[k, 4])
            print(line)
        tprev_end = last
    uo = (upd[:, 2] > 0) & (upd[:, 3] > 0)
    if uo.any():
        t0 = upd[uo, 0].min()
        en = np.sort((upd[uo, 2] - t0) / 100.0)
        print("   end times (every 16th WG, sorted):", " ".join(f"{x:.0f}" for x in en[::16]), f"last {en[-1]:.0f}")
        byidx = (upd[:, 2] - t0) / 100.0
        print("   end times by WG index (every 16th):", " ".join(f"{byidx[k]:.0f}" if uo[k] else "-" for k in range(0, 256, 16)))
        t0 = upd[uo, 0].min()
        e = (upd[uo, 0] - t0) / 100.0; tb = (upd[uo, 1] - upd[uo, 0]) / 100.0; en = (upd[uo, 2] - t0) / 100.0
        print(f" k_update of the block: {uo.sum()} WGs, entry {e.min():.1f}..{e.max():.1f} us, first tables after {tb.min():.1f}..{tb.max():.1f} (mean {tb.mean():.1f}), "
              f"spans {upd[uo, 3].min()}..{upd[uo, 3].max()}, table time per WG mean {upd[uo, 4].mean() / 100.0:.1f} max {upd[uo, 4].max() / 100.0:.1f}, "
              f"end {en.min():.1f}..{en.max():.1f} (mean {en.mean():.1f})")
    wv = np.zeros((5, 16), dtype=np.uint64)
    assert lib.gf2bv_probe_read_wave(wv.ctypes.data_as(ctypes.c_void_p)) == 0
    wv = wv.astype(np.int64)
    print(" k_update table build of workgroup 8 (us since its entry): parameters/prow %.1f, pivot rows staged %.1f, pass 0 %.1f, pass 1 %.1f"
          % tuple((wv[4, k] - upd[8, 0]) / 100.0 for k in range(4)))
    gj = np.zeros(4, dtype=np.uint64)
    assert lib.gf2bv_probe_read_gj(gj.ctypes.data_as(ctypes.c_void_p)) == 0
    gj = gj.astype(np.int64)
    print(" column-wise first chunk of unit 0 (last panel of the run): transpose in %.2f us, 64 pivots %.2f, transpose back + hand-over %.2f"
          % ((gj[1] - gj[0]) / 100.0, (gj[2] - gj[1]) / 100.0, (gj[3] - gj[2]) / 100.0))
buf.free()
