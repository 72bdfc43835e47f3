"""PRNG models for the recovery configurations (BASELINE configs[2], configs[4], the NLFSR example).  HARNESS, not product: the
reference ships such models in its package (gf2bv/crypto); here they live with the tests, written from the published algorithms, and
run unchanged on ints and on symbolic words (gf2bv_amd.BitVec / PackedBitVec) -- bench.py, examples/ and tests/ import them from here."""
from .lfsr import FibonacciLFSR, GaloisLFSR
from .mt19937 import MT19937
from .xoshiro import Xoshiro256starstar

__all__ = ["FibonacciLFSR", "GaloisLFSR", "MT19937", "Xoshiro256starstar"]
