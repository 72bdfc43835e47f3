"""In-tree build of the native pieces (hipcc for gfx950, g++ for the CPython shim).

    python gf2bv_amd/build.py            # or __graft_entry__.build()
(run it as a script: `python -m gf2bv_amd.build` would import the package, i.e. the extension it builds)

Outputs land next to this file so they travel with the source tree:
  libgf2bv_hip.so                       HIP kernels + C ABI (include/gf2bv_hip.h)
  _internal.cpython-3xx-*.so            the `gf2bv._internal`-compatible extension
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
HIP_LIB = os.path.join(HERE, "libgf2bv_hip.so")
EXT = os.path.join(HERE, "_internal" + sysconfig.get_config_var("EXT_SUFFIX"))


def source_id(sources: list) -> str:
    """sha256 over the sources' names and contents, first 16 hex digits: what `gf2bv_build_id()` of a library built from them says"""
    h = hashlib.sha256()
    for path in sources:
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def binary_id(target: str, marker: bytes):
    """the id compiled into an existing binary (read from the file, nothing is loaded), or None"""
    try:
        with open(target, "rb") as f:
            blob = f.read()
    except OSError:
        return None
    at = blob.find(marker)
    if at < 0:
        return None
    end = blob.find(b"\0", at)
    return blob[at + len(marker):end].decode(errors="replace")


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.sep not in cand or os.path.exists(cand)):
            return cand
    return "hipcc"


HIP_MARK = b"GF2BV_BUILD_ID="
EXT_MARK = b"GF2BV_SHIM_ID="


def build(force: bool = False, verbose: bool = False) -> dict:
    """Compile what is stale -- by CONTENT: the sha256 of the sources is compiled into each binary (`gf2bv_build_id()`,
    `_internal.build_id()`), and a binary is rebuilt when the id inside it is not the id of the sources next to it (round 5's
    mtime test could not say whether the binary on the GPU box was built from the tree that travelled with it).
    Returns and prints {"hip": ("built" | "reused", id), "shim": (...)}."""
    hip_src = [os.path.join(CSRC, "gf2_solver.hip"), os.path.join(CSRC, "gf2_kernels.hip.h"),
               os.path.join(HERE, "..", "include", "gf2bv_hip.h")]
    what = {}
    hid = source_id(hip_src)
    if force or binary_id(HIP_LIB, HIP_MARK) != hid:
        cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", f'-DGF2BV_BUILD_ID="{hid}"',
               hip_src[0], "-o", HIP_LIB]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        what["hip"] = ("built", hid)
    else:
        what["hip"] = ("reused", hid)
    ext_src = [os.path.join(CSRC, "_internal.cpp"), hip_src[2]]
    eid = source_id(ext_src) + "-" + hid
    if force or what["hip"][0] == "built" or binary_id(EXT, EXT_MARK) != eid:
        cmd = ["g++", "-O2", "-std=c++17", "-pthread", "-shared", "-fPIC", "-I" + sysconfig.get_paths()["include"],
               f'-DGF2BV_SHIM_ID="{eid}"', ext_src[0], "-L" + HERE, "-lgf2bv_hip", "-Wl,-rpath,$ORIGIN", "-o", EXT]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        what["shim"] = ("built", eid)
    else:
        what["shim"] = ("reused", eid)
    print(f"[gf2bv build] libgf2bv_hip.so {what['hip'][0]} {what['hip'][1]}; _internal {what['shim'][0]} {what['shim'][1]}", flush=True)
    return what


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
