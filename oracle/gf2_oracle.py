"""ctypes front-end of the CPU oracle.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module (see the header of oracle/gf2_oracle.c).  The product package gf2bv_amd never does.

It restates, on the CPU, the reference boundary ``gf2bv._internal.m4ri_solve`` and the
``AffineSpace`` enumeration orders (reference file:line relative to /root/reference):

* ``eqs_to_aug``      gf2bv/_internal.c:398-426  (matrix assembly; sign ignored, bits above
                      cols ignored, bit 0 = affine term -> RHS column)
* ``m4ri_solve``      gf2bv/_internal.c:359-502  (argument checks + solve + kernel)
* ``OracleSpace``     gf2bv/_internal.c:179-304  (dimension/origin/basis/get/__iter__)
* Gray / binary enumeration order  gf2bv/_internal.c:101-122 / :63-91

PARITY PIN STATUS: see gf2_oracle.c -- unique-solution answers are pinned by
examples/mt.py:38; rank-deficient tie-breaking follows M4RI's documented PLUQ contract
and is "parity unpinned" against a live M4RI.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libgf2oracle.so")
    src = os.path.join(_HERE, "gf2_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libgf2oracle.so", "libgf2m4ritime.so"])
    return so


def m4ri_time(n: int) -> dict:
    """True M4RI (mzd_pluq + mzd_pluq_solve_left on an n x n mzd_randomize matrix) timed through libgf2m4ritime.so, when the box has
    a libm4ri.so; {"found": False} otherwise (every box so far).  bench.py's cpu_baseline leg reports it beside the port."""
    so = os.path.join(_HERE, "libgf2m4ritime.so")
    src = os.path.join(_HERE, "m4ri_timing.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libgf2m4ritime.so"])
    L = ctypes.CDLL(so)
    sp, ss, rk = ctypes.c_double(0), ctypes.c_double(0), ctypes.c_int(0)
    name = ctypes.create_string_buffer(128)
    rc = L.gf2o_m4ri_time(ctypes.c_int(n), ctypes.byref(sp), ctypes.byref(ss), ctypes.byref(rk), name, 128)
    if rc != 0:
        return {"found": False, "why": "no libm4ri.so on this host" if rc == 1 else "libm4ri.so lacks an expected symbol"}
    return {"found": True, "library": name.value.decode(), "n": n, "seconds_pluq": sp.value, "seconds_solve_left": ss.value, "rank": rk.value,
            "input": "mzd_randomize (M4RI's generator), not the synthetic generator: a timing, not a parity pin"}


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        p64 = ctypes.POINTER(ctypes.c_uint64)
        L.gf2o_solve.restype = ctypes.c_void_p
        L.gf2o_solve.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                 ctypes.c_int, ctypes.c_int]
        for name, res in (("gf2o_status", ctypes.c_int), ("gf2o_rank", ctypes.c_int64),
                          ("gf2o_dim", ctypes.c_int64), ("gf2o_row_xors", ctypes.c_double),
                          ("gf2o_sweep_words", ctypes.c_double)):
            f = getattr(L, name)
            f.restype = res
            f.argtypes = [ctypes.c_void_p]
        for name in ("gf2o_pivcols", "gf2o_origin", "gf2o_basis"):
            f = getattr(L, name)
            f.restype = None
            f.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.gf2o_free.restype = None
        L.gf2o_free.argtypes = [ctypes.c_void_p]
        L.gf2o_max_threads.restype = ctypes.c_int
        L.gf2o_set_threads.argtypes = [ctypes.c_int]
        L.gf2o_gen_synthetic.restype = None
        L.gf2o_gen_synthetic.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                         ctypes.c_uint64]
        L.gf2o_planted_solution.restype = None
        L.gf2o_planted_solution.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint64]
        L.gf2o_check_solution.restype = ctypes.c_int64
        L.gf2o_check_solution.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                          ctypes.c_void_p]
        L.gf2o_synth_word.restype = ctypes.c_uint64
        L.gf2o_synth_word.argtypes = [ctypes.c_uint64, ctypes.c_int64, ctypes.c_int64]
        _ = p64
        _LIB = L
    return _LIB


# ----------------------------------------------------------------------------------------
# layout helpers

def words_for(cols: int) -> int:
    """64-bit words of an augmented row: columns 0..cols-1 plus the RHS column `cols`."""
    return (cols + 1 + 63) // 64


def eqs_to_aug(eqs, cols: int, stride: int | None = None) -> np.ndarray:
    """gf2bv/_internal.c:398-426: bit 0 of eq -> RHS (column `cols`), bit k -> column k-1."""
    wt = words_for(cols)
    stride = wt if stride is None else stride
    mask = (1 << (cols + 1)) - 1
    nbytes = stride * 8
    buf = bytearray(len(eqs) * nbytes)
    for r, e in enumerate(eqs):
        v = abs(int(e)) & mask            # CPython digits are sign-magnitude; sign is ignored
        row = (v >> 1) | ((v & 1) << cols)
        buf[r * nbytes:(r + 1) * nbytes] = row.to_bytes(nbytes, "little")
    return np.frombuffer(bytes(buf), dtype=np.uint64).reshape(len(eqs), stride).copy()


def words_to_int(words: np.ndarray) -> int:
    return int.from_bytes(np.ascontiguousarray(words, dtype=np.uint64).tobytes(), "little")


def int_to_words(v: int, nwords: int) -> np.ndarray:
    return np.frombuffer(int(v).to_bytes(nwords * 8, "little"), dtype=np.uint64).copy()


# ----------------------------------------------------------------------------------------
# solve

def solve_words(aug: np.ndarray, rows: int, cols: int, mode: int = 0, algo: int = 1) -> dict:
    """Run the CPU restatement on a packed augmented matrix (rows x stride uint64)."""
    L = lib()
    aug = np.ascontiguousarray(aug, dtype=np.uint64)
    stride = aug.shape[1] if aug.ndim == 2 else words_for(cols)
    h = L.gf2o_solve(aug.ctypes.data, rows, cols, stride, mode, algo)
    if not h:
        raise ValueError("gf2o_solve rejected its arguments")
    try:
        cw = (cols + 63) // 64
        rank, dim, status = L.gf2o_rank(h), L.gf2o_dim(h), L.gf2o_status(h)
        piv = np.zeros(max(rank, 1), dtype=np.int32)
        L.gf2o_pivcols(h, piv.ctypes.data)
        origin = np.zeros(max(cw, 1), dtype=np.uint64)
        L.gf2o_origin(h, origin.ctypes.data)
        basis = np.zeros((dim, cw), dtype=np.uint64)
        if mode == 1 and status == 0 and dim:
            L.gf2o_basis(h, basis.ctypes.data)
        return {"status": status, "rank": rank, "dim": dim, "pivcols": piv[:rank].copy(),
                "origin": origin[:cw], "basis": basis, "row_xors": L.gf2o_row_xors(h),
                "sweep_words": L.gf2o_sweep_words(h)}
    finally:
        L.gf2o_free(h)


class OracleSpace:
    """Restates AffineSpace (gf2bv/_internal.c:179-304) over Python ints."""

    def __init__(self, origin: int, basis: tuple[int, ...]):
        self.origin = origin
        self.basis = tuple(basis)

    @property
    def dimension(self) -> int:
        return len(self.basis)

    def get(self, n: int) -> int:          # _internal.c:242-273: binary, low `dimension` bits of |n|
        v = self.origin
        n = abs(n)
        for i, b in enumerate(self.basis):
            if (n >> i) & 1:
                v ^= b
        return v

    def __iter__(self):
        d = len(self.basis)
        if d <= 64:                        # _internal.c:101-122: reflected Gray code
            cur, idx = self.origin, 0
            while True:
                yield cur
                x = idx ^ (idx >> 1)
                idx = (idx + 1) & 0xFFFFFFFFFFFFFFFF
                y = idx ^ (idx >> 1)
                diff = ((x ^ y) & -(x ^ y)).bit_length() - 1 if (x ^ y) else 64
                if diff >= d or (d == 64 and idx == 0):
                    return
                cur ^= self.basis[diff]
        else:                              # _internal.c:63-91: binary counter, basis[0] = LSB
            state = [0] * (d + 1)
            while not state[d]:
                v = self.origin
                for r in range(d):
                    if state[r]:
                        v ^= self.basis[r]
                sentinel = 1
                for r in range(d):
                    state[r] ^= 1
                    if state[r]:
                        sentinel = 0
                        break
                state[d] = sentinel
                yield v


def m4ri_solve(eqs, cols, mode, algo: int = 1):
    """CPU restatement of gf2bv._internal.m4ri_solve (gf2bv/_internal.c:359-502)."""
    if not isinstance(eqs, list):
        raise TypeError("The first argument equations must be a list")
    if cols <= 0:
        raise ValueError("Number of columns must be positive")
    if mode not in (0, 1):
        raise ValueError("Invalid mode")
    if len(eqs) < cols:
        raise ValueError("Number of rows must be greater than or equal to number of columns, try pad with zeros.")
    for e in eqs:
        if not isinstance(e, int):
            raise TypeError("List items must be integers")
    aug = eqs_to_aug(eqs, cols)
    res = solve_words(aug, len(eqs), cols, mode, algo)
    if res["status"] != 0:
        return None
    origin = words_to_int(res["origin"])
    if mode == 0:
        return origin
    return OracleSpace(origin, tuple(words_to_int(b) for b in res["basis"]))


def brute_force_solutions(eqs, cols):
    """All x in [0, 2^cols) with every equation satisfied (bit 0 = constant).  Tiny cols only."""
    out = []
    for x in range(1 << cols):
        v = (x << 1) | 1
        if all(bin(e & v).count("1") % 2 == 0 for e in eqs):
            out.append(x)
    return out


# ----------------------------------------------------------------------------------------
# synthetic systems

def gen_synthetic(rows: int, cols: int, seed: int, stride: int | None = None) -> np.ndarray:
    stride = words_for(cols) if stride is None else stride
    aug = np.zeros((rows, stride), dtype=np.uint64)
    lib().gf2o_gen_synthetic(aug.ctypes.data, rows, cols, stride, seed)
    return aug


def planted_solution(cols: int, seed: int) -> np.ndarray:
    x = np.zeros((cols + 63) // 64, dtype=np.uint64)
    lib().gf2o_planted_solution(x.ctypes.data, cols, seed)
    return x


def check_solution(aug: np.ndarray, rows: int, cols: int, x: np.ndarray) -> int:
    aug = np.ascontiguousarray(aug, dtype=np.uint64)
    x = np.ascontiguousarray(x, dtype=np.uint64)
    return int(lib().gf2o_check_solution(aug.ctypes.data, rows, cols, aug.shape[1], x.ctypes.data))
