"""In-tree build of the native pieces (hipcc for gfx950, g++ for the CPython shim).

    python gf2bv_amd/build.py            # or __graft_entry__.build()
(run it as a script: `python -m gf2bv_amd.build` would import the package, i.e. the extension it builds)

Outputs land next to this file so they travel with the source tree:
  libgf2bv_hip.so                       HIP kernels + C ABI (include/gf2bv_hip.h)
  _internal.cpython-3xx-*.so            the `gf2bv._internal`-compatible extension
"""
from __future__ import annotations

import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
HIP_LIB = os.path.join(HERE, "libgf2bv_hip.so")
EXT = os.path.join(HERE, "_internal" + sysconfig.get_config_var("EXT_SUFFIX"))


def _stale(target: str, sources: list) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.sep not in cand or os.path.exists(cand)):
            return cand
    return "hipcc"


def build(force: bool = False, verbose: bool = False) -> None:
    hip_src = [os.path.join(CSRC, "gf2_solver.hip"), os.path.join(CSRC, "gf2_kernels.hip.h"),
               os.path.join(HERE, "..", "include", "gf2bv_hip.h")]
    if force or _stale(HIP_LIB, hip_src):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
               hip_src[0], "-o", HIP_LIB]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    ext_src = [os.path.join(CSRC, "_internal.cpp"), hip_src[2]]
    if force or _stale(EXT, ext_src + [HIP_LIB]):
        cmd = ["g++", "-O2", "-std=c++17", "-pthread", "-shared", "-fPIC", "-I" + sysconfig.get_paths()["include"],
               ext_src[0], "-L" + HERE, "-lgf2bv_hip", "-Wl,-rpath,$ORIGIN", "-o", EXT]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
