"""One device-resident solve_one of a large synthetic system with the residual check (default 524288 x 524288: 32 GiB input +
32 GiB working copy).  usage: largest_run.py [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gf2bv_amd import hip
n = int(sys.argv[1]) if len(sys.argv) > 1 else 524288
stride = hip.padded_stride(n)
buf = hip.DeviceBuffer(n * stride * 8)
hip.synth_device(buf.ptr, n, n, stride, 1234)
t = time.time()
sol = hip.solve_device(buf.ptr, n, n, stride, 0, time_kernels=True)
dt = time.time() - t
bad = hip.residual_device(buf.ptr, n, n, stride, sol.origin)
s = sol.stats
rate = 16 * s["sweep_words"] / (s["ms_sweep"] * 1e-3) / 1e12
print(f"N={n} rank={sol.rank} wall={dt:.2f}s eliminate={s['ms_eliminate'] / 1e3:.2f}s residual_rows={bad} bulk update {rate:.2f} TB/s per pass")
buf.free()
