"""PRNG models that run unchanged on ints and on BitVecs (harness for the MT19937 and
xoshiro256** recovery configurations; public algorithms, written fresh)."""
from .mt import MT19937, MersenneTwister
from .xoshiro import Xoshiro256starstar

__all__ = ["MT19937", "MersenneTwister", "Xoshiro256starstar"]
