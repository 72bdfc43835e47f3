"""PRNG models that run unchanged on ints and on BitVecs (harness for the MT19937,
xoshiro256** and NLFSR recovery configurations; public algorithms, written fresh)."""
from .lfsr import FibonacciLFSR, GaloisLFSR
from .mt import MT19937, MersenneTwister
from .xoshiro import Xoshiro256starstar

__all__ = ["FibonacciLFSR", "GaloisLFSR", "MT19937", "MersenneTwister", "Xoshiro256starstar"]
