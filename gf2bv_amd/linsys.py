"""LinearSystem: the solve front-end that feeds the MI355X solver.

Re-statement (own code) of gf2bv/__init__.py:137-287 of maple3142/gf2bv: same public
surface (``gens``, ``get_eqs``, ``solve_one``, ``solve_all``, ``solve_raw_*``,
``evaluate``, ``convert_sol``, pickling) and the same conventions -- but
``_internal.m4ri_solve`` is the HIP path in libgf2bv_hip.so, not M4RI.
"""
from __future__ import annotations

from typing import Iterable, Optional, Sequence, Union

from ._internal import AffineSpace, eqs_to_sage_mat_helper, m4ri_solve, m4ri_solve_many, mul_bit_quad
from .bitvec import BitVec

Zeros = Sequence[Union[BitVec, int]]


class DimensionTooLargeError(Exception):
    """Raised by solve_all when the solution space has more than 2**max_dimension points."""

    def __init__(self, message: str, space: AffineSpace):
        super().__init__(message)
        self.space = space


class LinearSystem:
    def __init__(self, sizes: Iterable[int]):
        self._sizes = list(sizes)
        self._cols = sum(self._sizes)
        # bit 0 of an equation int is the constant term; unknown g is bit g+1
        self._basis = [1 << i for i in range(self._cols + 1)]
        gens, at = [], 1
        for width in self._sizes:
            gens.append(BitVec(tuple(self._basis[at:at + width])))
            at += width
        self._vars = tuple(gens)

    def gens(self):
        return self._vars

    def __reduce__(self):
        return (self.__class__, (self._sizes,))

    # -- zeros -> equation ints (reference :214-227) ------------------------------------------------
    def get_eqs(self, zeros: Zeros) -> list:
        flat: list = []
        for z in zeros:
            if isinstance(z, BitVec):
                flat.extend(z._bits)
            else:
                flat.append(z)
        return [e for e in flat if e]          # literal zeros carry no information

    # -- Sage export (reference :167-212) --------------------------------------------------------------------
    def get_sage_mat_slow(self, zeros: Zeros, *, tqdm=lambda x, desc: x):
        """(A, b) over GF(2) with A x = b, as Sage objects -- needs Sage at call time (reference :167-192)."""
        from sage.all import GF, matrix, vector          # noqa: PLC0415  (optional dependency, as in the reference)
        eqs = self.get_eqs(zeros)
        F2 = GF(2)
        b = vector(F2, [e & 1 for e in eqs])
        A = matrix(F2, len(eqs), self._cols)
        for i, e in enumerate(tqdm(eqs, desc="Converting equations")):
            e >>= 1
            while e:
                low = e & -e
                A[i, low.bit_length() - 1] = 1
                e ^= low
        return A, b

    def get_sage_mat(self, zeros: Zeros):
        """(A, b) as Sage objects, the fast way (reference :194-212): the coefficient matrix travels as a two-colour PNG
        (`eqs_to_sage_mat_helper`, _internal.c:678-765 -- written without libgd here) into Sage's own unpickler for
        GF(2) matrices; the affine bits come back as a list of bools.  Needs Sage at call time."""
        import struct                                     # noqa: PLC0415

        from sage.all import GF, vector                   # noqa: PLC0415  (optional dependency, as in the reference)
        from sage.matrix.matrix_mod2_dense import unpickle_matrix_mod2_dense_v2      # noqa: PLC0415

        eqs = self.get_eqs(zeros)
        buf, affine = eqs_to_sage_mat_helper(eqs, self._cols)
        b = vector(GF(2), affine)
        signed = struct.unpack(f">{len(buf)}b", buf)      # Sage wants the PNG as signed chars
        return unpickle_matrix_mod2_dense_v2(len(eqs), self._cols, signed, len(signed), False), b

    # -- boundary call (reference :229-240) ------------------------------------------------------------
    def _solve_internal(self, zeros: Zeros, mode: int):
        eqs = self.get_eqs(zeros)
        if 1 in eqs:                            # the equation "1 = 0"
            return None
        if len(eqs) < self._cols:               # the boundary wants rows >= cols
            eqs.extend([0] * (self._cols - len(eqs)))
        return m4ri_solve(eqs, self._cols, mode)

    # -- raw int -> per-variable ints (reference :242-251) ----------------------------------------------
    def _convert_sol(self, s: int) -> tuple:
        parts = []
        for width in self._sizes:
            parts.append(s & ((1 << width) - 1))
            s >>= width
        assert s == 0, "Invalid solution"
        return tuple(parts)

    def convert_sol(self, s: int) -> Optional[tuple]:
        return self._convert_sol(s)

    def solve_raw_one(self, zeros: Zeros):
        return self._solve_internal(zeros, 0)

    def solve_raw_space(self, zeros: Zeros):
        return self._solve_internal(zeros, 1)

    def solve_all(self, zeros: Zeros, *, max_dimension: int = 16):
        space = self.solve_raw_space(zeros)
        if space is None:
            return
        if space.dimension > max_dimension:
            raise DimensionTooLargeError(
                f"Solution space (dim {space.dimension}) is too large, try increase max_dimension "
                f"({max_dimension}) if you want (there will be 2**dim solutions)",
                space=space,
            )
        for raw in space:
            sol = self.convert_sol(raw)
            if sol is not None:
                yield sol

    def solve_one(self, zeros: Zeros):
        raw = self._solve_internal(zeros, 0)
        return None if raw is None else self.convert_sol(raw)

    # -- batches (no counterpart in the reference) ----------------------------------------------------------
    # Independent instances of the same LinearSystem (one zeros list per instance / per output bit) go to
    # the GPU as ONE call: _internal.m4ri_solve_many -> gf2bv_solve_batch_digits, lock-step gangs.
    # Element i of the result is exactly what the single-system method returns for zeros_list[i].
    # `devices`: None = the module's default device (set_default_device / GF2BV_DEVICE, like solve_one), "all" = every visible
    # GPU, an int, or a sequence of device indices (the systems are sharded in contiguous blocks over them inside the library,
    # one host thread per entry, no collective -- gf2bv_solve_batch_digits_multi).
    def _solve_internal_many(self, zeros_list: Sequence[Zeros], mode: int, devices=None) -> list:
        eqs_list = [self.get_eqs(z) for z in zeros_list]
        live = [i for i, eqs in enumerate(eqs_list) if 1 not in eqs]      # "1 = 0" is decided on the host
        rows = max([self._cols] + [len(eqs_list[i]) for i in live])
        for i in live:
            eqs_list[i].extend([0] * (rows - len(eqs_list[i])))
        out: list = [None] * len(eqs_list)
        if live:
            for i, res in zip(live, m4ri_solve_many([eqs_list[i] for i in live], self._cols, mode, devices)):
                out[i] = res
        return out

    def solve_raw_one_many(self, zeros_list: Sequence[Zeros], devices=None) -> list:
        return self._solve_internal_many(zeros_list, 0, devices)

    def solve_raw_space_many(self, zeros_list: Sequence[Zeros], devices=None) -> list:
        return self._solve_internal_many(zeros_list, 1, devices)

    def solve_one_many(self, zeros_list: Sequence[Zeros], devices=None) -> list:
        return [None if raw is None else self.convert_sol(raw)
                for raw in self._solve_internal_many(zeros_list, 0, devices)]

    def evaluate(self, bv: BitVec, sol: tuple) -> int:
        raw, shift = 0, 0
        for value, width in zip(sol, self._sizes):
            raw |= value << shift
            shift += width
        return bv.evaluate(raw)


class QuadraticSystem(LinearSystem):
    """Quadratic equations over GF(2) by linearisation: on top of the n unknowns of ``sizes`` every product
    x_i x_j (j < i) is an unknown of its own, n(n-1)/2 of them after the linear ones, and the linearised system
    goes through the same solve path (e.g. 128 unknowns -> 8256 columns, the NLFSR example).  Solutions whose
    "product" unknowns do not equal the products of their linear part are filtered out by ``convert_sol``.

    Own restatement of gf2bv/__init__.py:290-408 (same surface: ``gens``, ``mul_bit``, ``bit_assert``,
    ``convert_sol``, ``solve_one`` = first of ``solve_all``, ``evaluate``, pickling).  One convention of the
    reference is kept on purpose: ``mul_bit`` combines the constant / linear parts of its operands as
    ``a & b`` (x_i^2 = x_i), i.e. the constant x linear cross terms are not formed -- callers multiply
    bits without constant terms (gf2bv/__init__.py:334-338)."""

    def __init__(self, sizes: Iterable[int]):
        sizes = list(sizes)
        n = sum(sizes)
        pairs = n * (n - 1) // 2
        super().__init__(sizes + [pairs])
        self._quad_sizes = sizes
        self._lin_size = n
        self._quad_size = pairs
        self._const_lin_mask = (1 << (n + 1)) - 1          # constant bit + the n linear unknowns

    def gens(self):
        return super().gens()[:-1]                          # the block of product unknowns is internal

    def __reduce__(self):
        return (self.__class__, (self._quad_sizes,))

    # product of two single-bit expressions, as an equation int over the linearised unknowns
    def _mul_bit(self, a: int, b: int) -> int:
        low = (a & self._const_lin_mask) & b
        # pair (i, j), j < i, sits at bit 1 + n + i(i-1)/2 + j; it is present iff a_i b_j + a_j b_i = 1
        return mul_bit_quad(self._lin_size, a >> 1, b >> 1, low, self._basis)

    def mul_bit(self, a: BitVec, b: BitVec) -> BitVec:
        if len(a) != 1 or len(b) != 1:
            raise ValueError("The inputs should be single bits")
        return BitVec((self._mul_bit(a._bits[0], b._bits[0]),))

    # "bit a equals v" plus everything that follows from it after multiplying by each unknown
    def _bit_assert(self, a: int, v: int) -> list:
        assert v in (0, 1), "Invalid bit"
        assert a not in (0, 1), "a should not be a constant"
        # DEVIATION from the reference, on purpose and documented: gf2bv/__init__.py:349 asserts
        # `a >> self._lin_size == 0`, which rejects every expression that contains the LAST linear unknown (bit
        # lin_size of an equation int: bit 0 is the constant, unknown g is bit g + 1) although it is a linear term;
        # the bound that matches the representation is lin_size + 1.  Every input the reference accepts gives the
        # same zeros here; inputs it rejects by that off-by-one are accepted.
        assert a >> (self._lin_size + 1) == 0, "Not a linear term"
        zeros = [a ^ v]
        for i in range(1, self._lin_size + 1):
            x = self._basis[i]
            if x == a:
                continue
            zeros.append(self._mul_bit(a, x) ^ (x if v else 0))     # a * x = v * x
        return zeros

    def bit_assert(self, a: BitVec, v: int) -> Zeros:
        if len(a) != 1:
            raise ValueError("The input should be a single bit")
        return self._bit_assert(a._bits[0], v)

    def _products_match(self, lin: int, quad: int) -> bool:
        n = self._lin_size
        for i in range(n):
            if (lin >> i) & 1:
                # products with x_i = 1: pairs (i, j) for j < i must repeat the low i bits of lin
                want = lin & ((1 << i) - 1)
            else:
                want = 0
            if quad & ((1 << i) - 1) != want:
                return False
            quad >>= i
        assert quad == 0, "Invalid quadratic part"
        return True

    def convert_sol(self, s: int) -> Optional[tuple]:
        lin = s & ((1 << self._lin_size) - 1)
        quad = s >> self._lin_size
        assert quad >> self._quad_size == 0, "Invalid solution"
        if not self._products_match(lin, quad):
            return None
        return self._convert_sol(lin)[:-1]

    def solve_one(self, zeros: Zeros):
        # the particular solution of the linearised system need not be consistent: take the first one that is
        for sol in self.solve_all(zeros):
            return sol
        return None

    def evaluate(self, bv: BitVec, sol: tuple) -> int:
        raw, shift = 0, 0
        for value, width in zip(sol, self._quad_sizes):
            raw |= value << shift
            shift += width
        return bv.evaluate(raw)
