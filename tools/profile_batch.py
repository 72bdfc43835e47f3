"""One gang batch (for rocprofv3): nsys synthetic n x n systems through gf2bv_solve_batch_device.  usage: profile_batch.py n nsys [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gf2bv_amd import hip
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
nsys = int(sys.argv[2]) if len(sys.argv) > 2 else 28
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
stride = hip.padded_stride(n)
buf = hip.DeviceBuffer(nsys * n * stride * 8)
for i in range(nsys):
    hip.synth_device(buf.ptr + i * n * stride * 8, n, n, stride, 5000 + i)
for r in range(reps):
    t = time.perf_counter()
    sols = hip.solve_batch_device(buf.ptr, nsys, n * stride, n, n, stride, 0, time_kernels=True)
    dt = time.perf_counter() - t
    print(f"{nsys} x {n}^2: {dt * 1e3:.1f} ms = {dt / nsys * 1e3:.2f} ms per system; gang {sols[0].stats['gang_systems']}, "
          f"gang eliminate {sols[0].stats['ms_eliminate']:.1f} ms, gang sweep {sols[0].stats['ms_sweep']:.1f} ms", flush=True)
buf.free()
