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
    hip.synth_device(buf.ptr, n, n, stride, int(os.environ.get("SEED", "1234")))
    t = time.time()
    sol = hip.solve_device(buf.ptr, n, n, stride, 0, time_kernels=tk)
    dt = time.time() - t
    s = sol.stats
    res = hip.residual_device(buf.ptr, n, n, stride, sol.origin) if os.environ.get("RESIDUAL", "1") == "1" else -1
    print(f"N={n} rank={sol.rank} wall={dt*1e3:.1f}ms elim={s['ms_eliminate']:.1f} sweep={s['ms_sweep']:.1f} back={s['ms_backsub']:.1f} total={s['ms_total']:.1f} fast_blocks={s['fast_blocks']} handovers={s['search_handovers']}"
          f" outer_blocks={s['outer_blocks']}"
          f" elim_TBs={s['sweep_words'] * 16 / max(s['ms_eliminate'], 1e-9) / 1e9:.3f} residual_rows={res}", flush=True)
buf.free()
