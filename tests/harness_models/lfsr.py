"""Right-shifting n-bit linear feedback shift registers in their two textbook forms (Golomb, "Shift Register Sequences"), over a
concrete int or a symbolic n-bit word.  Calling the register clocks it once and returns the bit that left it (the old bit 0).

  Galois    : the leaving bit is XORed into the tapped positions       state' = (state >> 1) ^ (taps if out else 0)
  Fibonacci : the parity of the tapped positions enters at the top     state' = (state >> 1) | (parity(state & taps) << (n - 1))

Harness for examples/nlfsr_recovery.py and the quadratic golden fixtures."""
from __future__ import annotations


class _Register:
    def __init__(self, n: int, mask: int, state):
        self.n, self.mask = n, mask & ((1 << n) - 1)
        self.state = state & ((1 << n) - 1)

    def _next(self, out):
        raise NotImplementedError

    def __call__(self):
        out = self.state & 1
        self.state = self._next(out)
        return out


class GaloisLFSR(_Register):
    def _next(self, out):
        shifted = self.state >> 1
        if isinstance(out, int):
            return shifted ^ (self.mask if out else 0)
        return shifted ^ (out.broadcast(0, len(self.state)) & self.mask)       # the leaving bit spread over the word, as data


class FibonacciLFSR(_Register):
    def _next(self, out):
        tapped = self.state & self.mask
        if isinstance(tapped, int):
            return (self.state >> 1) | ((bin(tapped).count("1") & 1) << (self.n - 1))
        top = tapped.sum().zeroext(self.n - 1) << (self.n - 1)                  # the parity bit, moved to position n - 1
        return (self.state >> 1) ^ top
