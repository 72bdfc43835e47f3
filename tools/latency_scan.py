"""Warm wall time of solve_one on device-resident synthetic systems across sizes (where does a small solve's time go?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gf2bv_amd import hip
for n in (int(a) for a in (sys.argv[1:] or "256 1024 2048 4096 8192 16384".split())):
    stride = hip.padded_stride(n)
    buf = hip.DeviceBuffer(n * stride * 8)
    hip.synth_device(buf.ptr, n, n, stride, 1234)
    best, s = 1e9, None
    for r in range(6):
        t = time.perf_counter(); sol = hip.solve_device(buf.ptr, n, n, stride, 0); dt = time.perf_counter() - t
        if dt < best: best, s = dt, sol.stats
    print(f"N={n:6d} wall {best*1e3:7.3f} ms  eliminate {s['ms_eliminate']:.3f} backsub {s['ms_backsub']:.3f} export {s['ms_export']:.3f} host_total {s['ms_total']:.3f} fast_blocks {s['fast_blocks']} rank {sol.rank}", flush=True)
    buf.free()
