"""xoshiro256** (Blackman & Vigna), polymorphic over ``int`` and BitVec state words.

The state update is GF(2)-linear; the ``**`` output scrambler is not, but it is a bijection
of the pre-update ``s1``, so the harness inverts it on the concrete side
(``step() ^ untemper(observed)``), as the reference's model does
(maple3142/gf2bv, gf2bv/crypto/xoshiro.py:24-50; examples/xoshiro.py:12).
"""
from __future__ import annotations

import secrets

from ..bitvec import BitVec

_M64 = (1 << 64) - 1
_INV5 = pow(5, -1, 1 << 64)
_INV9 = pow(9, -1, 1 << 64)


def _rotl(x, n):
    if isinstance(x, BitVec):
        return x.rotl(n)
    return ((x << n) | (x >> (64 - n))) & _M64


class Xoshiro256starstar:
    def __init__(self, s):
        if len(s) != 4:
            raise ValueError("invalid state")
        self.s = list(s)

    @staticmethod
    def generate():
        return Xoshiro256starstar([secrets.randbits(64) for _ in range(4)])

    @staticmethod
    def temper(s1: int) -> int:
        return (_rotl((s1 * 5) & _M64, 7) * 9) & _M64

    @staticmethod
    def untemper(out: int) -> int:
        return (_rotl((out * _INV9) & _M64, 64 - 7) * _INV5) & _M64

    def step(self):
        """Advance the state; returns the PRE-update s1 (what the scrambler is applied to)."""
        s0, s1, s2, s3 = self.s
        picked = s1
        t = (s1 << 17) & _M64
        s2 = s2 ^ s0
        s3 = s3 ^ s1
        s1 = s1 ^ s2
        s0 = s0 ^ s3
        s2 = s2 ^ t
        s3 = _rotl(s3, 45)
        self.s = [s0, s1, s2, s3]
        return picked

    def __call__(self) -> int:
        return self.temper(self.step())
