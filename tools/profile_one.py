"""One device-resident synthetic solve (for rocprofv3).  usage: profile_one.py N [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gf2bv_amd import hip
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
tk = os.environ.get("TIME_KERNELS", "0") == "1"
stride = hip.padded_stride(n)
buf = hip.DeviceBuffer(n * stride * 8)
for r in range(reps):
    hip.synth_device(buf.ptr, n, n, stride, 1234)
    t = time.time()
    sol = hip.solve_device(buf.ptr, n, n, stride, 0, time_kernels=tk)
    dt = time.time() - t
    s = sol.stats
    print(f"N={n} rank={sol.rank} wall={dt*1e3:.1f}ms elim={s['ms_eliminate']:.1f} sweep={s['ms_sweep']:.1f} back={s['ms_backsub']:.1f} total={s['ms_total']:.1f} fast_blocks={s['fast_blocks']} handovers={s['search_handovers']}", flush=True)
buf.free()
