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
