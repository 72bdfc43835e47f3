"""gf2bv_amd -- MI355X-native GF(2) linear-system solver behind gf2bv's API.

Drop-in for the solve path of maple3142/gf2bv: ``LinearSystem`` / ``gens()`` /
``solve_one`` / ``solve_all`` and the ``_internal`` boundary (``m4ri_solve``,
``AffineSpace``) keep their meaning; the bit-packed matrix assembly and the elimination /
back-substitution / kernel basis run as HIP kernels on gfx950 (gf2bv_amd/csrc).
There is no CPU fallback: importing works anywhere, solving needs the GPU.
"""
from ._internal import AffineSpace, eqs_to_sage_mat_helper, m4ri_solve, mul_bit_quad, to_bits, tuple_where, xor_tuple
from .bitvec import BitVec
from .linsys import DimensionTooLargeError, LinearSystem, QuadraticSystem, Zeros


def __getattr__(name):
    # the packed front-end needs numpy (>= 1.20; np.bitwise_count of 2.x is used when present): imported on first use so
    # that the package itself has no import-time dependency beyond the standard library
    if name in ("PackedBitVec", "PackedLinearSystem", "packed"):
        import importlib
        mod = importlib.import_module(".packed", __name__)
        return mod if name == "packed" else getattr(mod, name)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")


__all__ = [
    "AffineSpace", "BitVec", "DimensionTooLargeError", "LinearSystem", "PackedBitVec", "PackedLinearSystem",
    "QuadraticSystem", "Zeros",
    "eqs_to_sage_mat_helper", "m4ri_solve", "mul_bit_quad", "to_bits", "tuple_where", "xor_tuple",
]
