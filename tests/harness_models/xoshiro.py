"""xoshiro256** (Blackman & Vigna, "Scrambled linear pseudorandom number generators", ACM TOMS 47(4), 2021; reference C code
xoshiro256starstar.c, public domain) over concrete or symbolic 64-bit words.

The state transition is GF(2)-linear; the `**` scrambler -- rotl(s1 * 5, 7) * 9 -- is a bijection of s1, so a recovery harness
undoes it on the observed output and equates the result with the symbolic PRE-update s1: `step() ^ unscramble(observed)`."""
from __future__ import annotations

import secrets

MASK = (1 << 64) - 1


def rotl64(x, n: int):
    return ((x << n) | (x >> (64 - n))) & MASK if isinstance(x, int) else x.rotl(n)


class Xoshiro256starstar:
    def __init__(self, s):
        words = list(s)
        if len(words) != 4:
            raise ValueError("xoshiro256 has four state words")
        self.s = words

    @classmethod
    def generate(cls):
        return cls(secrets.randbits(64) for _ in range(4))

    @staticmethod
    def scramble(s1: int) -> int:
        return rotl64(s1 * 5 & MASK, 7) * 9 & MASK

    @staticmethod
    def unscramble(out: int) -> int:
        return rotl64(out * pow(9, -1, 1 << 64) & MASK, 57) * pow(5, -1, 1 << 64) & MASK

    temper, untemper = scramble, unscramble          # (the names the reference's examples use)

    def step(self):
        """one state transition; returns s1 as it was BEFORE it (the scrambler's input)"""
        a, b, c, d = self.s
        c, d = c ^ a, d ^ b                            # s2 ^= s0; s3 ^= s1
        self.s = [a ^ d, b ^ c, c ^ ((b << 17) & MASK), rotl64(d, 45)]
        return b

    def __call__(self) -> int:
        return self.scramble(self.step())
