"""Scan synthetic-generator seeds until the N x N matrix has full rank (P ~ 0.289 per seed)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gf2bv_amd import hip
for n in [int(a) for a in sys.argv[1:]] or [65536]:
    stride = hip.padded_stride(n)
    buf = hip.DeviceBuffer(n * stride * 8)
    for seed in range(1234, 1334):
        hip.synth_device(buf.ptr, n, n, stride, seed)
        sol = hip.solve_device(buf.ptr, n, n, stride, 0)
        print(n, seed, sol.rank, flush=True)
        if sol.rank == n:
            print("FULL", n, seed, flush=True)
            break
    buf.free()
