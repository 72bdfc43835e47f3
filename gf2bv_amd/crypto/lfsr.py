"""Linear feedback shift registers in their two standard forms, running unchanged on ints and on BitVecs
(harness for the NLFSR recovery example; public algorithms, written fresh; counterpart of the reference's
gf2bv/crypto/lfsr.py)."""
from __future__ import annotations

from ..bitvec import BitVec


class GaloisLFSR:
    """Shift right, output the bit that falls out, XOR the tap mask in when it is 1."""

    def __init__(self, n: int, mask: int, state):
        full = (1 << n) - 1
        self.n = n
        self.mask = mask & full
        self.state = state & full

    def __call__(self):
        out = self.state & 1
        self.state = self.state >> 1
        if isinstance(out, BitVec):
            self.state = self.state ^ (out.broadcast(0, len(self.state)) & self.mask)
        elif out:
            self.state ^= self.mask
        return out


class FibonacciLFSR:
    """Shift right, output the bit that falls out, feed the parity of the tapped bits in at the top."""

    def __init__(self, n: int, mask: int, state):
        full = (1 << n) - 1
        self.n = n
        self.mask = mask & full
        self.state = state & full

    def __call__(self):
        out = self.state & 1
        if isinstance(self.state, BitVec):
            fb = (self.state & self.mask).sum()                           # 1-bit BitVec
            self.state = (self.state >> 1) ^ (fb.zeroext(self.n - 1) << (self.n - 1))
        else:
            fb = bin(self.state & self.mask).count("1") & 1
            self.state = (self.state >> 1) | (fb << (self.n - 1))
        return out
