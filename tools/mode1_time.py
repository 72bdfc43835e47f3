import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gf2bv_amd import hip
for n in (8192, 32768, 65536):
    stride = hip.padded_stride(n)
    buf = hip.DeviceBuffer(n * stride * 8)
    hip.synth_device(buf.ptr, n, n, stride, 5000)
    for mode in (0, 1):
        for rep in range(2):
            t = time.time(); s = hip.solve_device(buf.ptr, n, n, stride, mode); dt = time.time() - t
        print(n, "mode", mode, "rank", s.rank, "dim", s.dimension, f"{dt*1e3:.1f} ms", {k: round(v, 2) for k, v in s.stats.items() if k.startswith("ms_")})
    buf.free()
