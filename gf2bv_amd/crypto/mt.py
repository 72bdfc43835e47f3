"""Mersenne Twister, polymorphic over ``int`` and :class:`gf2bv_amd.BitVec` state words.

Public algorithm (Matsumoto & Nishimura 1998).  Same constructor / ``getrandbits`` surface as
the reference model (maple3142/gf2bv, gf2bv/crypto/mt.py:6-107) so examples/mt.py-style
scripts run unchanged: ``MT19937(lin.gens()).getrandbits(k) ^ observed`` builds the zeros list.
"""
from __future__ import annotations

import random

from ..bitvec import BitVec


class MersenneTwister:
    def __init__(self, mt, w, n, m, r, a, u, d, s, b, t, c, l):  # noqa: E741
        if len(mt) != n:
            raise ValueError("invalid parameters")
        full = (1 << w) - 1
        if max(r, u, s, t, l) > w or max(a, b, c, d) > full:
            raise ValueError("invalid parameters")
        self.mt = list(mt)
        self.w, self.n, self.m, self.r = w, n, m, r
        self.a, self.u, self.d, self.s, self.b, self.t, self.c, self.l = a, u, d, s, b, t, c, l
        self.w1 = full
        self.lmsk = (1 << r) - 1            # low r bits
        self.umsk = full ^ self.lmsk        # high w-r bits
        self.mti = n                        # state is "exhausted": first output twists

    def twist(self):
        n, m, st = self.n, self.m, self.mt
        for i in range(n):
            y = (st[i] & self.umsk) ^ (st[(i + 1) % n] & self.lmsk)
            if isinstance(y, BitVec):
                mag = y.broadcast(0, self.w) & self.a      # (y & 1) ? a : 0, symbolically
            else:
                mag = self.a if (y & 1) else 0
            st[i] = st[(i + m) % n] ^ (y >> 1) ^ mag

    def temper(self, y):
        y = y ^ ((y >> self.u) & self.d)
        y = y ^ ((y << self.s) & self.w1 & self.b)
        y = y ^ ((y << self.t) & self.w1 & self.c)
        return y ^ (y >> self.l)

    def __call__(self):
        if self.mti >= self.n:
            self.twist()
            self.mti = 0
        y = self.mt[self.mti]
        self.mti += 1
        return self.temper(y)

    def _top_bits(self, k):
        out = self()
        if isinstance(out, BitVec):
            return out[self.w - k:]
        return out >> (self.w - k)

    def getrandbits(self, k=None):
        """CPython's ``random.getrandbits``: top k bits of each word, words little-endian."""
        if k is None:
            k = self.w
        if k < 0:
            raise ValueError("number of bits cannot be negative")
        if k == 0:
            return 0
        if k <= self.w:
            return self._top_bits(k)
        acc = 0
        nwords = (k + self.w - 1) // self.w
        for i in range(nwords):
            part = self._top_bits(min(k, self.w))
            acc = acc | (part.lshift_ext(self.w * i) if isinstance(part, BitVec) else part << (self.w * i))
            k -= self.w
        return acc


class MT19937(MersenneTwister):
    """The 32-bit generator behind CPython's ``random`` module."""

    def __init__(self, mt):
        super().__init__(mt, 32, 624, 397, 31, 0x9908B0DF, 11, 0xFFFFFFFF, 7, 0x9D2C5680,
                         15, 0xEFC60000, 18)

    def to_python_random(self):
        r = random.Random(0)
        r.setstate((3, (*self.mt, self.mti), None))
        return r
