"""Throughput of independent systems on one GPU (BASELINE configs[3] per-GPU share)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gf2bv_amd import hip
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
nsys = int(sys.argv[2]) if len(sys.argv) > 2 else 8
stride = hip.padded_stride(n)
buf = hip.DeviceBuffer(nsys * n * stride * 8)
for i in range(nsys):
    hip.synth_device(buf.ptr + i * n * stride * 8, n, n, stride, 5000 + i)
for rep in range(2):
    t = time.time()
    sols = hip.solve_batch_device(buf.ptr, nsys, n * stride, n, n, stride, 0)
    dt = time.time() - t
    bad = sum(hip.residual_device(buf.ptr + i * n * stride * 8, n, n, stride, s.origin) for i, s in enumerate(sols))
    print(f"batch n={n} nsys={nsys}: {dt*1e3:.1f} ms total, {dt*1e3/nsys:.1f} ms/system, residual rows {bad}, ranks {[s.rank for s in sols][:4]}", flush=True)
t = time.time()
s = hip.solve_device(buf.ptr, n, n, stride, 0)
print(f"single: {(time.time()-t)*1e3:.1f} ms")
