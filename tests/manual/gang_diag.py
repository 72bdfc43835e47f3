import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from gf2bv_amd import hip
n, nsys = 32768, 24
stride = hip.padded_stride(n)
buf = hip.DeviceBuffer(nsys * n * stride * 8)
for i in range(nsys):
    hip.synth_device(buf.ptr + i * n * stride * 8, n, n, stride, 5000 + i)
for rep in range(int(os.environ.get("REPS", "6"))):
    sols = hip.solve_batch_device(buf.ptr, nsys, n * stride, n, n, stride, 0)
    bad = [hip.residual_device(buf.ptr + i * n * stride * 8, n, n, stride, s.origin) for i, s in enumerate(sols)]
    print("rep", rep, "bad systems:", [(i, b, sols[i].rank, sols[i].status) for i, b in enumerate(bad) if b], flush=True)
