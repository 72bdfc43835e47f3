"""The two restatements of the reference's solve path must agree: oracle/gf2_oracle (the result contract S1-S5 read off
the reduced row echelon form) and oracle/m4ri_mimic (the same calls gf2bv/_internal.c:428-489 makes, walked through
operationally with explicit P / Q transposition arrays).  10^4 random systems, most of them rank deficient: same
consistency verdict, same origin, same kernel basis IN THE SAME ORDER."""
import random

from oracle import gf2_oracle as O
from oracle import m4ri_mimic as M


def _system(rng):
    cols = rng.randint(1, 22)
    rows = cols + rng.randint(0, 6)
    kind = rng.random()
    dens = rng.choice((0.1, 0.3, 0.5, 0.8))
    rank_cap = rng.randint(0, cols) if kind < 0.7 else None
    plant = rng.getrandbits(cols)
    eqs = []
    for i in range(rows):
        if rank_cap is not None and i >= rank_cap and rank_cap > 0 and rng.random() < 0.9:
            a = 0
            for _ in range(rng.randint(1, 3)):
                a ^= eqs[rng.randrange(rank_cap)] >> 1
        elif rank_cap == 0:
            a = 0
        else:
            a = sum((rng.random() < dens) << j for j in range(cols))
        eqs.append((a << 1) | (bin(a & plant).count("1") & 1))
    if rng.random() < 0.15:                       # some inconsistent ones
        eqs[rng.randrange(rows)] ^= 1
    if rng.random() < 0.1:                        # homogeneous: the "all affine bits zero" shortcut (_internal.c:451-454)
        eqs = [e & ~1 for e in eqs]
    rng.shuffle(eqs)
    return eqs, cols


def test_operational_walk_equals_contract_shortcut():
    rng = random.Random(20260928)
    seen_none = seen_deficient = 0
    for _ in range(10000):
        eqs, cols = _system(rng)
        want = O.m4ri_solve(list(eqs), cols, 1)
        got = M.m4ri_solve(list(eqs), cols, 1)
        if want is None:
            assert got is None
            seen_none += 1
            continue
        assert got is not None
        origin, basis = got
        assert origin == want.origin and basis == want.basis, (eqs, cols)
        assert M.m4ri_solve(list(eqs), cols, 0) == O.m4ri_solve(list(eqs), cols, 0) == origin
        seen_deficient += len(basis) > 0
    assert seen_none > 300 and seen_deficient > 5000


def test_known_answers_through_the_walk():
    # README 4-variable system (SURVEY 8a-S hand KAT) and a full-rank one
    assert M.m4ri_solve([15, 20, 11, 0], 4, 1) == (0b0001, (0b0101,))
    assert M.m4ri_solve([15, 20, 11, 0], 4, 0) == 0b0001
    assert M.m4ri_solve([0b11, 0b100], 2, 1) == (0b01, ())
    assert M.m4ri_solve([0b1, 0b0], 2, 1) is None
