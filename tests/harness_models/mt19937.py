"""MT19937 (Matsumoto & Nishimura, "Mersenne Twister: a 623-dimensionally equidistributed uniform pseudo-random number generator",
ACM TOMACS 8(1), 1998) with CPython's `getrandbits` framing (Modules/_randommodule.c: the TOP k bits of each 32-bit output, words
little-endian), over concrete ints or symbolic 32-bit words.

One flat generator, written from the paper's recurrence:
    x[k + 624] = x[k + 397] ^ ((x[k] & UPPER | x[k + 1] & LOWER) >> 1) ^ (A if the low bit of that mix is set)
    y ^= y >> 11;  y ^= (y << 7) & B;  y ^= (y << 15) & C;  y ^= y >> 18
A symbolic word w only needs `^ & >> <<`, `w[i:]` (drop the low i bits), `w.broadcast(0, 32)` (bit 0 repeated) and
`w.lshift_ext(n)` (shift without truncation) -- what gf2bv_amd.BitVec and PackedBitVec offer."""
from __future__ import annotations

import random

WORD, N, M = 32, 624, 397
A = 0x9908B0DF
UPPER, LOWER = 0x80000000, 0x7FFFFFFF
B, C = 0x9D2C5680, 0xEFC60000
FULL = 0xFFFFFFFF


def _is_int(x) -> bool:
    return isinstance(x, int)


class MT19937:
    def __init__(self, state):
        words = list(state)
        if len(words) != N:
            raise ValueError(f"MT19937 has {N} state words, got {len(words)}")
        self.mt = words
        self.mti = N                       # CPython's name for the read position; N = regenerate before the next output

    def _regenerate(self):
        x = self.mt
        for k in range(N):
            mix = (x[k] & UPPER) ^ (x[(k + 1) % N] & LOWER)
            # the conditional XOR of A, as data: bit 0 of `mix` spread over the word
            sel = (A if mix & 1 else 0) if _is_int(mix) else mix.broadcast(0, WORD) & A
            x[k] = x[(k + M) % N] ^ (mix >> 1) ^ sel
        self.mti = 0

    def genrand(self):
        """the next tempered 32-bit output"""
        if self.mti >= N:
            self._regenerate()
        y = self.mt[self.mti]
        self.mti += 1
        y = y ^ (y >> 11)
        y = y ^ ((y << 7) & FULL & B)
        y = y ^ ((y << 15) & FULL & C)
        return y ^ (y >> 18)

    __call__ = genrand

    def _high(self, k: int):
        y = self.genrand()
        return y >> (WORD - k) if _is_int(y) else y[WORD - k:]

    def getrandbits(self, k: int = WORD):
        if k < 0:
            raise ValueError("number of bits cannot be negative")
        if k == 0:
            return 0
        if k <= WORD:
            return self._high(k)
        out, at = 0, 0
        while at < k:                      # full words first, the last one takes what is left: k - at bits
            part = self._high(min(WORD, k - at))
            out = out | (part << at if _is_int(part) else part.lshift_ext(at))
            at += WORD
        return out

    def to_python_random(self) -> random.Random:
        """a `random.Random` continuing from this (concrete) state"""
        r = random.Random(0)
        r.setstate((3, tuple(self.mt) + (self.mti,), None))
        return r
