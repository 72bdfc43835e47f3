"""Phase times inside k_block_fast (the one-launch block search) for chosen blocks; builds the -DGF2_STEP_PROBE library.
    python tools/probe_fast.py [N] [block ...]"""
import ctypes, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "_probe", "libgf2bv_hip_probe.so")
SRC = os.path.join(ROOT, "gf2bv_amd", "csrc", "gf2_solver.hip")
if not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(SRC), os.path.getmtime(SRC.replace("gf2_solver.hip", "gf2_kernels.hip.h"))):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-DGF2_STEP_PROBE", SRC, "-o", LIB])
if "--build-only" in sys.argv:
    sys.exit(0)
os.environ["GF2BV_LIB"] = LIB
sys.path.insert(0, ROOT)
import numpy as np
from gf2bv_amd import hip
args = [a for a in sys.argv[1:] if not a.startswith("-")]
n = int(args[0]) if args else 65536
blocks = [int(a) for a in args[1:]] or [8, n // 512, n // 256 - 16]
lib = hip.lib(); stride = hip.padded_stride(n)
buf = hip.DeviceBuffer(n * stride * 8)
hip.synth_device(buf.ptr, n, n, stride, 1234); hip.solve_device(buf.ptr, n, n, stride, 0)
names = (["entry", "candidates in LDS"] + [f"panel {g}: {x}" for g in range(4) for x in ("search done", "pivot rows formed", "candidates narrowed")]
         + ["published", "chain tail: inputs ready", "handed over", "end: before the release", "after"])
for b in blocks:
    assert lib.gf2bv_probe_set(ctypes.c_int(b * 4)) == 0
    hip.solve_device(buf.ptr, n, n, stride, 0)
    w = np.zeros(32, dtype=np.uint64)
    assert lib.gf2bv_probe_read_fast(w.ctypes.data_as(ctypes.c_void_p)) == 0
    w = w.astype(np.int64)
    print(f"N={n} block {b}: " + "  ".join(f"{nm} {(w[k] - w[0]) / 100.0:.1f}" for k, nm in enumerate(names) if w[k]))
