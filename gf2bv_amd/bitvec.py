"""Symbolic bit-vectors over GF(2).

Re-statement (own code) of the reference's ``BitVec`` (maple3142/gf2bv,
gf2bv/__init__.py:21-134).  A symbolic bit is a Python int: bit 0 is the constant term, bit
``i+1`` the coefficient of unknown ``i``.  A ``BitVec`` is a tuple of such ints, least
significant bit first.  Running ordinary bit-twiddling code on BitVecs produces the ``zeros``
list that :class:`gf2bv_amd.LinearSystem` hands to the MI355X solver.  Nothing here is
accelerated; it is the producer side of the hot path.
"""
from __future__ import annotations

from ._internal import to_bits, tuple_where, xor_tuple


def _parity(v: int) -> int:
    return bin(v).count("1") & 1


class BitVec:
    __slots__ = ("_bits",)

    def __init__(self, bits: tuple):
        self._bits = bits                      # LSB first

    # -- container protocol (reference :29-37) ------------------------------------------------
    def __len__(self):
        return len(self._bits)

    def __getitem__(self, key):
        picked = self._bits[key]
        # a single index still yields a 1-bit BitVec so that bv[0] ^ bv cannot be written by accident
        return BitVec(picked if isinstance(key, slice) else (picked,))

    # -- xor (reference :39-49) ----------------------------------------------------------------
    def __xor__(self, other):
        if isinstance(other, BitVec):
            if len(other._bits) != len(self._bits):
                raise ValueError("Cannot mix bitvecs of different lengths")
            rhs = other._bits
        else:
            rhs = to_bits(len(self._bits), other)
        return BitVec(xor_tuple(self._bits, rhs))

    __rxor__ = __xor__
    __pow__ = __xor__                          # alias, convenient under Sage's preparser

    # -- shifts / rotations (reference :51-62, :104-108) -----------------------------------------
    def __rshift__(self, n: int):
        return self if n == 0 else BitVec(self._bits[n:] + (0,) * n)

    def __lshift__(self, n: int):
        return self if n == 0 else BitVec((0,) * n + self._bits[:-n])

    def lshift_ext(self, n: int):
        return BitVec((0,) * n + self._bits)

    def rotr(self, n: int):
        return BitVec(self._bits[n:] + self._bits[:n])

    def rotl(self, n: int):
        return BitVec(self._bits[-n:] + self._bits[:-n])

    # -- masks (reference :64-102) -----------------------------------------------------------------
    def __and__(self, mask: int):
        sel = to_bits(len(self._bits), mask)
        if all(sel):
            return self
        return BitVec(tuple_where(sel, self._bits, 0))

    __rand__ = __and__

    def __or__(self, mask):
        if not isinstance(mask, BitVec):
            sel = to_bits(len(self._bits), mask)
            if all(sel):
                return BitVec(sel)
            return BitVec(tuple_where(sel, 1, self._bits))
        short, long_ = (self, mask) if len(self._bits) <= len(mask._bits) else (mask, self)
        merged = list(long_._bits)
        for i, (x, y) in enumerate(zip(short._bits, long_._bits)):
            if x not in (0, 1) and y not in (0, 1):
                raise ValueError("Cannot compute logical or using bitvecs with non-zero bits")
            if x == 1 or y == 1:
                merged[i] = 1
            elif x == 0:
                merged[i] = y
            else:
                merged[i] = x
        return BitVec(tuple(merged))

    __ror__ = __or__

    def __mod__(self, n: int):
        if n & (n - 1):
            raise ValueError("modulo non-power-of-2 is not a linear operation")
        return self & (n - 1)

    # -- reductions / reshaping (reference :110-126) ---------------------------------------------
    def sum(self):
        acc = 0
        for b in self._bits:
            acc ^= b
        return BitVec((acc,))

    def zeroext(self, n: int):
        return BitVec(self._bits + (0,) * n)

    def signext(self, n: int):
        return BitVec(self._bits + (self._bits[-1],) * n)

    def broadcast(self, i: int, n: int):
        return BitVec((self._bits[i],) * n)

    def dup(self, n: int):
        return BitVec(self._bits * n)

    def concat(self, other: "BitVec"):
        return BitVec(self._bits + other._bits)

    # -- evaluation (reference :128-134) -----------------------------------------------------------
    def evaluate(self, s: int) -> int:
        """Value of this BitVec under the raw solution ``s`` (bit j of s = unknown j)."""
        point = (s << 1) | 1
        value = 0
        for i, b in enumerate(self._bits):
            value |= _parity(b & point) << i
        return value
