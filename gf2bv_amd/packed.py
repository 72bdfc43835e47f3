"""Packed symbolic front-end: BitVec algebra on bit matrices instead of tuples of Python ints.

The reference represents a symbolic bit as a Python int (bit 0 = constant term, bit k = coefficient of unknown
k-1) and a BitVec as a tuple of such ints (gf2bv/__init__.py:21-134); every XOR of two 32-bit BitVecs over 19968
unknowns then touches 32 ints of 2.5 KB one by one through the interpreter (or the C helpers of
gf2bv/_internal.c:504-676), and the finished ``zeros`` are handed over as a list of ~20000 PyLongs that
``m4ri_solve`` reads bit by bit.  For the MT19937 recovery of examples/mt.py the front-end costs seconds, the solve
tens of milliseconds.

Here a ``PackedBitVec`` of n bits is ONE numpy array ``[n, W]`` of little-endian 64-bit words, W = ceil((cols+1)/64),
row i = symbolic bit i in the SAME bit order as the reference's ints.  XOR / shift / rotate / mask are whole-array
numpy operations; ``PackedLinearSystem`` stacks the rows of the ``zeros`` and gives the buffer straight to
``_internal.m4ri_solve_packed`` -- the device pack kernel reads it as 32-bit digits -- so no PyLong list exists at any
point.  Same surface and semantics as ``BitVec`` / ``LinearSystem`` (it IS a BitVec: the PRNG models in
tests.harness_models run on it unchanged); ``get_eqs`` still returns the reference's list of ints for callers that want it.
"""
from __future__ import annotations

import operator
from typing import Iterable, Optional, Sequence

import numpy as np

from ._internal import m4ri_solve_packed
from .bitvec import BitVec
from .linsys import DimensionTooLargeError


def _const_bits(n: int, value: int) -> np.ndarray:
    """low n bits of the MAGNITUDE of value, LSB first, as uint64 0/1 -- to_bits of the reference walks the digits of
    |a| (gf2bv/_internal.c:504-531), so a negative constant contributes abs(value), not its two's complement"""
    value = abs(int(value)) & ((1 << n) - 1)
    return np.array([(value >> i) & 1 for i in range(n)], dtype=np.uint64)


def _popcount64(a: np.ndarray) -> np.ndarray:
    if hasattr(np, "bitwise_count"):                   # numpy >= 2.0
        return np.bitwise_count(a)
    b = a.view(np.uint8).reshape(a.shape + (8,))       # numpy 1.x: per-byte table
    return _POP8[b].sum(axis=-1)


_POP8 = np.array([bin(i).count("1") for i in range(256)], dtype=np.uint64)


class PackedBitVec(BitVec):
    __slots__ = ("_rows",)

    def __init__(self, rows: np.ndarray):
        self._rows = rows                              # [nbits, W] uint64, treated as immutable

    # compatibility with code that reads the reference's representation
    @property
    def _bits(self) -> tuple:
        return tuple(int.from_bytes(r.tobytes(), "little") for r in self._rows)

    def _zero_rows(self, n: int) -> np.ndarray:
        return np.zeros((n, self._rows.shape[1]), dtype=np.uint64)

    def _coerce(self, other) -> np.ndarray:
        if isinstance(other, PackedBitVec):
            if other._rows.shape != self._rows.shape:
                raise ValueError("Cannot mix bitvecs of different lengths")
            return other._rows
        if isinstance(other, BitVec):
            raise TypeError("cannot mix packed and tuple-of-int BitVecs")
        rhs = self._zero_rows(len(self))
        rhs[:, 0] = _const_bits(len(self), other)      # a constant only has the affine bit
        return rhs

    # -- container protocol (reference :29-37) ------------------------------------------------
    def __len__(self):
        return self._rows.shape[0]

    def __getitem__(self, key):
        if isinstance(key, slice):
            return PackedBitVec(self._rows[key])
        i = operator.index(key)
        n = self._rows.shape[0]
        if not -n <= i < n:                            # (also what ends `for b in bv`: BitVec has no __iter__)
            raise IndexError("PackedBitVec index out of range")
        i %= n
        return PackedBitVec(self._rows[i:i + 1])

    # -- xor (reference :39-49) ----------------------------------------------------------------
    def __xor__(self, other):
        return PackedBitVec(self._rows ^ self._coerce(other))

    __rxor__ = __xor__
    __pow__ = __xor__

    # -- shifts / rotations (reference :51-62, :104-108) -----------------------------------------
    def __rshift__(self, n: int):
        if n == 0:
            return self
        return PackedBitVec(np.concatenate([self._rows[n:], self._zero_rows(n)]))      # (n > len grows it, as bits[n:] + (0,) * n does)

    def __lshift__(self, n: int):
        if n == 0:
            return self
        return PackedBitVec(np.concatenate([self._zero_rows(n), self._rows[:-n]]))

    def lshift_ext(self, n: int):
        return PackedBitVec(np.concatenate([self._zero_rows(n), self._rows]))

    def rotr(self, n: int):
        return PackedBitVec(np.roll(self._rows, -n, axis=0))

    def rotl(self, n: int):
        return PackedBitVec(np.roll(self._rows, n, axis=0))

    # -- masks (reference :64-102) -----------------------------------------------------------------
    def __and__(self, mask: int):
        if isinstance(mask, BitVec):
            raise TypeError("AND of two symbolic vectors is not linear (use QuadraticSystem)")
        sel = _const_bits(len(self), mask)
        if sel.all():
            return self
        return PackedBitVec(self._rows * sel[:, None])

    __rand__ = __and__

    def __or__(self, mask):
        if not isinstance(mask, BitVec):
            sel = _const_bits(len(self), mask)
            out = self._rows * (np.uint64(1) - sel)[:, None]
            out[:, 0] |= sel                           # selected bits become the constant 1
            return PackedBitVec(out)
        if not isinstance(mask, PackedBitVec):
            raise TypeError("cannot mix packed and tuple-of-int BitVecs")
        short, long_ = (self, mask) if len(self) <= len(mask) else (mask, self)
        merged = long_._rows.copy()
        one = np.zeros(self._rows.shape[1], dtype=np.uint64)
        one[0] = 1
        for i in range(len(short)):
            x, y = short._rows[i], long_._rows[i]
            xc = not x[1:].any() and x[0] <= 1         # a constant 0 / 1
            yc = not y[1:].any() and y[0] <= 1
            if not xc and not yc:
                raise ValueError("Cannot compute logical or using bitvecs with non-zero bits")
            if (xc and x[0] == 1) or (yc and y[0] == 1):
                merged[i] = one
            elif xc:                                   # x == 0
                merged[i] = y
            else:
                merged[i] = x
        return PackedBitVec(merged)

    __ror__ = __or__

    def __mod__(self, n: int):
        if n & (n - 1):
            raise ValueError("modulo non-power-of-2 is not a linear operation")
        return self & (n - 1)

    # -- reductions / reshaping (reference :110-126) ---------------------------------------------
    def sum(self):
        return PackedBitVec(np.bitwise_xor.reduce(self._rows, axis=0, keepdims=True))

    def zeroext(self, n: int):
        return PackedBitVec(np.concatenate([self._rows, self._zero_rows(n)]))

    def signext(self, n: int):
        return PackedBitVec(np.concatenate([self._rows, np.repeat(self._rows[-1:], n, axis=0)]))

    def broadcast(self, i: int, n: int):
        return PackedBitVec(np.repeat(self._rows[i:i + 1] if i != -1 else self._rows[-1:], n, axis=0))

    def dup(self, n: int):
        return PackedBitVec(np.tile(self._rows, (n, 1)))

    def concat(self, other: "PackedBitVec"):
        return PackedBitVec(np.concatenate([self._rows, other._rows]))

    # -- evaluation (reference :128-134) -----------------------------------------------------------
    def evaluate(self, s: int) -> int:
        W = self._rows.shape[1]
        point = np.frombuffer((((s << 1) | 1) & ((1 << (64 * W)) - 1)).to_bytes(8 * W, "little"), dtype=np.uint64)
        par = _popcount64(self._rows & point[None, :]).sum(axis=1) & 1
        return int(sum(int(b) << i for i, b in enumerate(par)))


class PackedLinearSystem:
    """LinearSystem (gf2bv/__init__.py:137-287) on PackedBitVecs: same methods, same results, no list of ints on the
    way to the solver."""

    def __init__(self, sizes: Iterable[int]):
        self._sizes = list(sizes)
        self._cols = sum(self._sizes)
        self._words = (self._cols + 1 + 63) // 64
        gens, at = [], 1                               # unknown g is bit g + 1 of an equation
        for width in self._sizes:
            rows = np.zeros((width, self._words), dtype=np.uint64)
            pos = np.arange(at, at + width)
            rows[np.arange(width), pos >> 6] = np.uint64(1) << (pos & 63).astype(np.uint64)
            gens.append(PackedBitVec(rows))
            at += width
        self._vars = tuple(gens)

    def gens(self):
        return self._vars

    def __reduce__(self):
        return (self.__class__, (self._sizes,))

    # -- zeros -> one [n, W] array (the packed twin of get_eqs, reference :214-227) ------------------
    def get_rows(self, zeros: Sequence) -> np.ndarray:
        """the stacked non-zero rows of ``zeros`` as a fresh [n, W] array (the solve methods use the shared buffer of
        ``_stack_rows`` instead and never keep it)"""
        return self._stack_rows(zeros).copy()

    def _stack_rows(self, zeros: Sequence) -> np.ndarray:
        parts = []
        for z in zeros:
            if isinstance(z, PackedBitVec):
                parts.append(z._rows)
            elif isinstance(z, BitVec):
                raise TypeError("cannot mix packed and tuple-of-int BitVecs")
            else:                                      # a bare equation int
                r = np.frombuffer((int(z) & ((1 << (64 * self._words)) - 1)).to_bytes(8 * self._words, "little"),
                                  dtype=np.uint64)
                parts.append(r[None, :])
        n = sum(len(x) for x in parts)
        # (stacked into a buffer that is kept between calls: a fresh 50 MB allocation is paid for in page faults, several
        # times the copy itself.  The returned array is a view of it, valid until the next call: internal use only.)
        buf = getattr(self, "_rowbuf", None)
        if buf is None or len(buf) < n:
            buf = self._rowbuf = np.empty((max(n, 1), self._words), dtype=np.uint64)
        rows = np.concatenate(parts, out=buf[:n]) if parts else buf[:0]
        nz = np.bitwise_or.reduce(rows, axis=1) != 0   # literal zeros carry no information
        return rows if nz.all() else rows[nz]          # (the usual case costs no second copy of the ~50 MB of an MT19937 system)

    def get_eqs(self, zeros: Sequence) -> list:
        """the reference's list of equation ints (for callers that want it; the solve methods do not build it)"""
        return [int.from_bytes(r.tobytes(), "little") for r in self._stack_rows(zeros)]

    # -- boundary call (reference :229-240) ------------------------------------------------------------
    def _solve_internal(self, zeros: Sequence, mode: int):
        rows = self._stack_rows(zeros)
        cand = np.flatnonzero(rows[:, 0] == 1) if len(rows) else ()                # the equation "1 = 0": word 0 is 1 ...
        if len(cand) and (~rows[cand, 1:].any(axis=1)).any():                        # ... and nothing else is set
            return None
        if len(rows) < self._cols:                     # the boundary wants rows >= cols
            rows = np.concatenate([rows, np.zeros((self._cols - len(rows), self._words), dtype=np.uint64)])
        rows = np.ascontiguousarray(rows)
        return m4ri_solve_packed(rows, rows.shape[0], self._words, self._cols, mode)

    def _convert_sol(self, s: int) -> tuple:
        parts = []
        for width in self._sizes:
            parts.append(s & ((1 << width) - 1))
            s >>= width
        assert s == 0, "Invalid solution"
        return tuple(parts)

    def convert_sol(self, s: int) -> Optional[tuple]:
        return self._convert_sol(s)

    def solve_raw_one(self, zeros: Sequence):
        return self._solve_internal(zeros, 0)

    def solve_raw_space(self, zeros: Sequence):
        return self._solve_internal(zeros, 1)

    def solve_one(self, zeros: Sequence):
        raw = self._solve_internal(zeros, 0)
        return None if raw is None else self.convert_sol(raw)

    def solve_all(self, zeros: Sequence, *, max_dimension: int = 16):
        space = self.solve_raw_space(zeros)
        if space is None:
            return
        if space.dimension > max_dimension:
            raise DimensionTooLargeError(
                f"Solution space (dim {space.dimension}) is too large, try increase max_dimension "
                f"({max_dimension}) if you want (there will be 2**dim solutions)",
                space=space,
            )
        for raw in space:
            sol = self.convert_sol(raw)
            if sol is not None:
                yield sol

    def evaluate(self, bv: BitVec, sol: tuple) -> int:
        raw, shift = 0, 0
        for value, width in zip(sol, self._sizes):
            raw |= value << shift
            shift += width
        return bv.evaluate(raw)
