"""The C-ABI shared library: loads, exports exactly what include/gf2bv_hip.h declares, validates
arguments like the reference boundary, and refuses to solve without a GPU.  No compute calls."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

from gf2bv_amd import hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "gf2bv_hip.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gf2bv_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_all_exported():
    decl = declared_symbols()
    assert decl == sorted(hip.EXPORTS)
    L = hip.lib()
    for name in decl:
        assert hasattr(L, name), name
    dyn = subprocess.run(["nm", "-D", "--defined-only", hip.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (gf2bv_[a-z_0-9]+)", dyn))
    assert set(decl) <= exported


def test_no_torch_types_in_abi():
    text = open(HEADER).read()
    assert "torch" not in text.lower().replace("no torch", "") and "at::" not in text and "hipStream_t" not in text.replace("hipStream_t or NULL", "")


def test_argument_validation_precedes_device_use():
    L = hip.lib()
    h = ctypes.c_void_p()
    aug = np.zeros((4, 1), dtype=np.uint64)
    assert L.gf2bv_solve_words(aug.ctypes.data, 4, 0, 1, 0, 0, ctypes.byref(h)) == 1          # cols <= 0
    assert b"columns must be positive" in L.gf2bv_last_error()
    assert L.gf2bv_solve_words(aug.ctypes.data, 4, 4, 1, 7, 0, ctypes.byref(h)) == 1          # bad mode
    assert b"Invalid mode" in L.gf2bv_last_error()
    assert L.gf2bv_solve_words(aug.ctypes.data, 3, 4, 1, 0, 0, ctypes.byref(h)) == 1          # rows < cols
    assert b"greater than or equal" in L.gf2bv_last_error()
    assert L.gf2bv_solve_words(aug.ctypes.data, 80, 70, 1, 0, 0, ctypes.byref(h)) == 1        # stride too small
    with pytest.raises(ValueError):
        hip.solve_words(aug, 3, 4)
    assert L.gf2bv_version() >= 100


def test_no_cpu_fallback():
    if hip.device_count() > 0:
        pytest.skip("a GPU is present")
    aug = np.zeros((4, 1), dtype=np.uint64)
    with pytest.raises(hip.HipError, match="no HIP device"):
        hip.solve_words(aug, 4, 4)
    with pytest.raises(hip.HipError):
        hip.DeviceBuffer(1024)


def test_space_combine_host_helper():
    L = hip.lib()
    origin = np.array([0b0001, 7], dtype=np.uint64)
    basis = np.array([[0b0101, 0], [0b1000, 1], [0, 1 << 63]], dtype=np.uint64)
    out = np.zeros(2, dtype=np.uint64)
    sel = np.array([0b101], dtype=np.uint64)
    L.gf2bv_space_combine(origin.ctypes.data, basis.ctypes.data, 3, 2, sel.ctypes.data, 1, out.ctypes.data)
    assert list(out) == [0b0001 ^ 0b0101, 7 ^ (1 << 63)]


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under gf2bv_amd/ may reference it, and outside tests/ only
    __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it."""
    pkg = os.path.join(ROOT, "gf2bv_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle" not in src.lower() or f == "__none__", os.path.join(dirpath, f)
    for sub in ("tools", "examples"):
        for f in os.listdir(os.path.join(ROOT, sub)):
            if f.endswith((".py", ".hip")):
                src = open(os.path.join(ROOT, sub, f), errors="replace").read()
                assert "import gf2_oracle" not in src and "from oracle" not in src, os.path.join(sub, f)
    bench = open(os.path.join(ROOT, "bench.py")).read()
    # every import of the oracle sits inside a cpu_baseline* function (the synthetic headline sample; the list-of-int systems
    # of the c3 / c5 legs): timed beside the GPU numbers and used as their checker, never as the thing measured
    parts = bench.split("from oracle import")
    assert 2 <= len(parts) <= 3
    for head in (("from oracle import".join(parts[:k])) for k in range(1, len(parts))):
        assert head.rsplit("\ndef ", 1)[-1].startswith("cpu_baseline"), head.rsplit("\ndef ", 1)[-1][:40]
    assert "oracle" not in bench.split("def timed_single")[1].split("def run_single")[0]      # the timed region itself


def test_round5_entry_points_without_a_device():
    """The planning / pool / staging entry points of round 5 on a box without a GPU: the gang planner is a pure function, the pool
    calls report "no device" instead of touching one, pinned staging is refused (the binding then falls back to malloc), and the
    optional true-M4RI timing says that there is no libm4ri here."""
    import ctypes
    L = hip.lib()
    # 64 systems of 32768^2 with a whole MI355X free: two gangs of 32 (one per host thread, multiples of 8 for a system per XCD)
    assert L.gf2bv_plan_gang(64, 32768, 32768, 280 * 10 ** 9) == 32
    assert L.gf2bv_plan_gang(512, 32768, 32768, 200 * 10 ** 9) == 32
    assert L.gf2bv_plan_gang(5, 2048, 2048, 10 ** 9) >= 1 and L.gf2bv_plan_gang(0, 1, 1, 1) == 0
    assert L.gf2bv_plan_gang(64, 32768, 32768, 2 * 10 ** 9) <= 6          # (little memory left: the gang shrinks to what fits)
    if hip.device_count() == 0:
        assert L.gf2bv_pool_trim(0) == -1
        p = ctypes.c_void_p()
        assert L.gf2bv_host_alloc(1 << 20, ctypes.byref(p)) == 2 and not p.value
    assert L.gf2bv_pool_idle_bytes(-1) == -1
    L.gf2bv_host_free(None)
    assert L.gf2bv_host_pool_trim() >= 0
    from oracle import gf2_oracle as O
    r = O.m4ri_time(256)
    assert r["found"] in (True, False) and (r["found"] or "libm4ri" in r["why"])
