"""Phase times inside k_block_sparse (round 5): builds the library with -DGF2_SPARSE_DEBUG into /tmp and runs one MT19937 variant.
usage: probe_sparse.py [bs]"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = "/tmp/libgf2_sparse_dbg.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-DGF2_SPARSE_DEBUG",
                       os.path.join(ROOT, "gf2bv_amd/csrc/gf2_solver.hip"), "-o", lib])
os.environ["GF2BV_LIB"] = lib
sys.path.insert(0, ROOT)
import runpy
from gf2bv_amd import hip
L = hip.lib()
buf = (ctypes.c_ulonglong * 8)()
for bs in [int(a) for a in sys.argv[1:]] or [32]:
    L.gf2bv_sparse_probe_read(buf, 1)
    sys.argv = ["mt_stats.py", str(bs)]
    runpy.run_path(os.path.join(ROOT, "tools/mt_stats.py"), run_name="__main__")
    L.gf2bv_sparse_probe_read(buf, 1)
    names = ["pool", "selection rounds", "gj_columns + owners", "tables + narrow", "publish", "first-64 attempt"]
    tot = sum(buf[i] for i in range(6)) or 1
    print(f"bs={bs}: k_block_sparse phases over the 2 solves: " + ", ".join(f"{n} {buf[i] / 100 / 154:.1f} us/block" for i, n in enumerate(names)) + f"; total {tot / 100 / 154:.1f} us/block", flush=True)
