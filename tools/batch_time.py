"""Throughput of independent systems on one GPU (BASELINE configs[3] per-GPU share).  usage: batch_time.py [n] [nsys] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gf2bv_amd import hip
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
nsys = int(sys.argv[2]) if len(sys.argv) > 2 else 8
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
stride = hip.padded_stride(n)
buf = hip.DeviceBuffer(nsys * n * stride * 8)
for i in range(nsys):
    hip.synth_device(buf.ptr + i * n * stride * 8, n, n, stride, 5000 + i)
times = []
for rep in range(reps):
    t = time.time()
    sols = hip.solve_batch_device(buf.ptr, nsys, n * stride, n, n, stride, 0, time_kernels=os.environ.get("TIME_KERNELS", "0") == "1")
    times.append(time.time() - t)
bad = sum(hip.residual_device(buf.ptr + i * n * stride * 8, n, n, stride, s.origin) for i, s in enumerate(sols))
best = min(times[1:] or times)
print(f"batch n={n} nsys={nsys}: " + " ".join(f"{t * 1e3:.0f}" for t in times) + f" ms; best {best * 1e3 / nsys:.3f} ms/system = {nsys / best:.0f} systems/s, "
      f"gang {sols[0].stats['gang_systems']}, residual rows {bad}, retries {max(s.stats['handover_retries'] for s in sols)}, outer blocks {sols[0].stats['outer_blocks']}", flush=True)
buf.free()
