"""Seeded system generators shared by the parity tests (inputs only -- no solver here)."""
from __future__ import annotations

import random


def random_system(rng: random.Random, rows: int, cols: int, density: float = 0.5, rank_cap: int | None = None,
                  consistent: bool = True, zero_rows: int = 0):
    """Equation ints (bit 0 = constant, bit k = coefficient of variable k-1), rows >= cols.

    rank_cap: rows beyond the first rank_cap are XOR-combinations of earlier ones.
    consistent: constants follow a planted solution; else a few constants are flipped
    (which makes the system inconsistent with high probability when rank-deficient rows exist).
    """
    plant = rng.getrandbits(cols)
    eqs = []
    for i in range(rows - zero_rows):
        if rank_cap is not None and i >= rank_cap and rank_cap > 0:
            a = 0
            for _ in range(rng.randint(1, 3)):
                a ^= eqs[rng.randrange(rank_cap)] >> 1
        elif density >= 0.5:
            a = rng.getrandbits(cols)
        else:
            a = 0
            for _ in range(max(1, int(density * cols))):
                a |= 1 << rng.randrange(cols)
        eqs.append((a << 1) | (bin(a & plant).count("1") & 1))
    eqs += [0] * zero_rows
    if not consistent:
        for _ in range(3):
            eqs[rng.randrange(rows - zero_rows)] ^= 1
    return eqs


def structured_system(rng: random.Random, rows: int, cols: int, kind: str):
    """Dense systems with a planted structure that steers the panel search into its special paths:
      'zero_cols'   a few all-zero columns inside otherwise dense panels (a panel stays a few pivots short for good)
      'dup_cols'    some columns are copies of earlier ones (dependent columns: same, but only after elimination)
      'dup_head'    the first 64 rows are copies of a handful of rows (the column-wise first chunk finds few pivots)
      'dead_head'   the first 40 rows are zero and the next 50 are copies of one row
    Returns equation ints consistent with a planted solution."""
    plant = rng.getrandbits(cols)
    keep = (1 << cols) - 1
    dup = []
    if kind == "zero_cols":
        for c in rng.sample(range(cols), min(cols // 7 + 1, 9)):
            keep &= ~(1 << c)
    elif kind == "dup_cols":
        dup = [(c, rng.randrange(c)) for c in rng.sample(range(1, cols), min(cols // 9 + 1, 7))]
    coeffs = []
    for i in range(rows):
        a = rng.getrandbits(cols) & keep
        for c, src in dup:
            a = (a & ~(1 << c)) | (((a >> src) & 1) << c)
        coeffs.append(a)
    if kind == "dup_head":
        base = coeffs[:5]
        for i in range(min(64, rows)):
            coeffs[i] = base[rng.randrange(5)] ^ (base[rng.randrange(5)] if rng.random() < 0.5 else 0)
    elif kind == "dead_head":
        for i in range(min(40, rows)):
            coeffs[i] = 0
        for i in range(40, min(90, rows)):
            coeffs[i] = coeffs[min(90, rows - 1)]
    return [(a << 1) | (bin(a & plant).count("1") & 1) for a in coeffs]
