"""Third oracle algorithm: the reference's solve path restated OPERATIONALLY.  TEST INFRASTRUCTURE ONLY.

gf2_oracle.{c,py} compute the result contract S1-S5 directly from the reduced row echelon form.  This
module instead walks through what gf2bv/_internal.c does with M4RI's objects, step by step, on small systems (rows are
Python ints, bit j = column j):

  * ``pluq``              _internal.c:431-433  _mzd_pluq(A, P, Q, 0) = PLE with LAPACK-style transposition arrays P
                          (rows) and Q (columns), L kept in compressed form, then mzd_apply_p_right_trans_tri on the
                          first r rows (pivot columns moved onto the diagonal: U = [U1 | U2], U1 unit upper triangular)
  * ``pluq_solve_left``   _internal.c:438-447  _mzd_pluq_solve_left(A, r, P, Q, B, 0, 1): mzd_apply_p_left(B, P),
                          forward substitution with L, the consistency check on rows r.., back substitution with U1,
                          free variables zeroed, mzd_apply_p_left_trans(B, Q)
  * ``kernel_left_pluq``  _internal.c:309-357  R[0:r] = U1^-1 U2, identity below, mzd_apply_p_left_trans(R, Q);
                          the basis is R transposed (_internal.c:486)
  * ``m4ri_solve``        _internal.c:428-489  the sequence of those calls, incl. the "all affine bits zero" shortcut
                          (:451-454)

M4RI itself (release 20260122, setup.py:14-17) is neither vendored in the reference tree nor installed, so the
permutation conventions are stated from its documented semantics:
  mzd_apply_p_left(A, P)        : for i ascending:  swap rows i, P[i]
  mzd_apply_p_left_trans(A, P)  : for i descending: swap rows i, P[i]
  mzd_apply_p_right_trans(A, Q) : for i ascending:  swap columns i, Q[i]
tests/test_oracle_mimic.py checks that this operational walk and the S1-S5 shortcut give the same origin, the same
basis in the same order and the same consistency verdict on 10^4 random rank-deficient systems.  That removes the
derivation risk inside S3/S4; it is still not a diff against a live M4RI ("parity unpinned" stays in force).
"""
from __future__ import annotations


def _swap_cols(row: int, a: int, b: int) -> int:
    if a == b:
        return row
    x = ((row >> a) ^ (row >> b)) & 1
    return row ^ ((x << a) | (x << b))


def pluq(A: list, nrows: int, ncols: int):
    """In place.  Returns (r, P, Q, L): after the call A[0:r] holds U (pivot i on the diagonal, column-permuted by Q),
    L[i] bit k = the multiplier of (permuted) row i with respect to pivot k (M4RI keeps it in the lower left corner)."""
    P, Q = list(range(nrows)), list(range(ncols))
    L = [0] * nrows
    r = 0
    for j in range(ncols):
        if r == nrows:
            break
        piv = next((i for i in range(r, nrows) if (A[i] >> j) & 1), None)
        if piv is None:
            continue
        P[r] = piv
        A[r], A[piv] = A[piv], A[r]
        L[r], L[piv] = L[piv], L[r]                  # a row swap carries the compressed L part along
        Q[r] = j
        for i in range(r + 1, nrows):
            if (A[i] >> j) & 1:
                A[i] ^= A[r]                          # (clears column j as well: E has zeros below its pivots)
                L[i] |= 1 << r
        r += 1
    # mzd_apply_p_right_trans_tri on the first r rows: column transpositions in ascending order
    for i in range(r):
        for k in range(r):
            A[k] = _swap_cols(A[k], i, Q[i])
    return r, P, Q, L


def pluq_solve_left(A: list, r: int, P: list, Q: list, L: list, B: list, nrows: int, ncols: int):
    """B: list of nrows bits (one right-hand side), in place.  Returns -1 when inconsistent (B is then garbage)."""
    for i in range(len(P)):                           # mzd_apply_p_left(B, P)
        B[i], B[P[i]] = B[P[i]], B[i]
    for i in range(r):                                # mzd_trsm_lower_left(L, Y1): unit lower triangular
        for k in range(i):
            if (L[i] >> k) & 1:
                B[i] ^= B[k]
    for i in range(r, nrows):                         # the check: Y2 + A2 * Y1 must vanish (A2 = rows r.. of L)
        acc = B[i]
        for k in range(r):
            if (L[i] >> k) & 1:
                acc ^= B[k]
        if acc:
            return -1
    for i in range(r - 1, -1, -1):                    # mzd_trsm_upper_left(U1, Y1)
        for k in range(i + 1, r):
            if (A[i] >> k) & 1:
                B[i] ^= B[k]
    for i in range(r, nrows):                         # free variables (and the surplus rows) set to zero
        B[i] = 0
    for i in range(ncols - 1, -1, -1):                # mzd_apply_p_left_trans(B, Q)
        B[i], B[Q[i]] = B[Q[i]], B[i]
    return 0


def kernel_left_pluq(A: list, r: int, Q: list, ncols: int):
    """Returns R as a list of ncols rows of (ncols - r) bits, or None when r == ncols."""
    if r == ncols:
        return None
    nk = ncols - r
    R = [0] * ncols
    for i in range(r):                                # RU = A[0:r, r:ncols] (mzd_read_bits / mzd_xor_bits loop)
        R[i] = (A[i] >> r) & ((1 << nk) - 1)
    for i in range(r - 1, -1, -1):                    # mzd_trsm_upper_left(U1, RU)
        for k in range(i + 1, r):
            if (A[i] >> k) & 1:
                R[i] ^= R[k]
    for i in range(nk):
        R[r + i] |= 1 << i
    for i in range(ncols - 1, -1, -1):                # mzd_apply_p_left_trans(R, Q)
        R[i], R[Q[i]] = R[Q[i]], R[i]
    return R


def m4ri_solve(eqs: list, cols: int, mode: int):
    """None | origin int | (origin int, basis tuple) following _internal.c:398-489 call by call."""
    rows = len(eqs)
    A = [(e >> 1) & ((1 << cols) - 1) for e in eqs]   # _internal.c:398-426
    B = [e & 1 for e in eqs]
    b_nonzero = any(B)
    r, P, Q, L = pluq(A, rows, cols)
    if b_nonzero:
        if pluq_solve_left(A, r, P, Q, L, B, rows, cols) != 0:
            return None
        origin = sum(B[j] << j for j in range(cols))
    else:
        origin = 0                                    # _internal.c:451-454
    if mode == 0:
        return origin
    R = kernel_left_pluq(A, r, Q, cols)
    if R is None:
        return origin, ()
    basis = tuple(sum(((R[j] >> i) & 1) << j for j in range(cols)) for i in range(cols - r))    # transpose, :486
    return origin, basis
