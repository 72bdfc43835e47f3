"""LinearSystem: the solve front-end that feeds the MI355X solver.

Re-statement (own code) of gf2bv/__init__.py:137-287 of maple3142/gf2bv: same public
surface (``gens``, ``get_eqs``, ``solve_one``, ``solve_all``, ``solve_raw_*``,
``evaluate``, ``convert_sol``, pickling) and the same conventions -- but
``_internal.m4ri_solve`` is the HIP path in libgf2bv_hip.so, not M4RI.
"""
from __future__ import annotations

from typing import Iterable, Optional, Sequence, Union

from ._internal import AffineSpace, m4ri_solve, m4ri_solve_many
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
    def _solve_internal_many(self, zeros_list: Sequence[Zeros], mode: int) -> list:
        eqs_list = [self.get_eqs(z) for z in zeros_list]
        live = [i for i, eqs in enumerate(eqs_list) if 1 not in eqs]      # "1 = 0" is decided on the host
        rows = max([self._cols] + [len(eqs_list[i]) for i in live])
        for i in live:
            eqs_list[i].extend([0] * (rows - len(eqs_list[i])))
        out: list = [None] * len(eqs_list)
        if live:
            for i, res in zip(live, m4ri_solve_many([eqs_list[i] for i in live], self._cols, mode)):
                out[i] = res
        return out

    def solve_raw_one_many(self, zeros_list: Sequence[Zeros]) -> list:
        return self._solve_internal_many(zeros_list, 0)

    def solve_raw_space_many(self, zeros_list: Sequence[Zeros]) -> list:
        return self._solve_internal_many(zeros_list, 1)

    def solve_one_many(self, zeros_list: Sequence[Zeros]) -> list:
        return [None if raw is None else self.convert_sol(raw) for raw in self._solve_internal_many(zeros_list, 0)]

    def evaluate(self, bv: BitVec, sol: tuple) -> int:
        raw, shift = 0, 0
        for value, width in zip(sol, self._sizes):
            raw |= value << shift
            shift += width
        return bv.evaluate(raw)
