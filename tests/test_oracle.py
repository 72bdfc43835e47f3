"""The CPU oracle against brute force, against itself (two algorithms), and against the
hand-derivable known answers.  No GPU."""
import random

import numpy as np
import pytest

from oracle import gf2_oracle as O
from tests.systems import random_system


def test_readme_kat():
    # m4ri_solve([15,20,11,0],4,1) -> origin 0b0001, basis (0b0101,)
    sp = O.m4ri_solve([15, 20, 11, 0], 4, 1)
    assert sp.origin == 0b0001 and sp.basis == (0b0101,) and sp.dimension == 1
    assert list(sp) == [0b0001, 0b0100]
    assert O.m4ri_solve([15, 20, 11, 0], 4, 0) == 0b0001
    assert sp.get(0) == 1 and sp.get(1) == 4 and sp.get(3) == 4      # only the low `dimension` bits count


@pytest.mark.parametrize("algo", [0, 1])
def test_vs_brute_force(algo):
    rng = random.Random(100 + algo)
    for _ in range(400):
        cols = rng.randint(1, 11)
        rows = rng.randint(cols, cols + 6)
        dens = rng.choice([0.1, 0.3, 0.5])
        eqs = [sum((rng.random() < dens) << k for k in range(cols + 1)) for _ in range(rows)]
        sp = O.m4ri_solve(eqs, cols, 1, algo=algo)
        bf = O.brute_force_solutions(eqs, cols)
        if sp is None:
            assert not bf
            assert O.m4ri_solve(eqs, cols, 0, algo=algo) is None
            continue
        sols = list(sp)
        assert sorted(sols) == sorted(bf) and len(set(sols)) == 2 ** sp.dimension
        assert O.m4ri_solve(eqs, cols, 0, algo=algo) == sp.origin
        # S3: origin has every free variable = 0; S4: basis[i] has exactly its own free variable set
        res = O.solve_words(O.eqs_to_aug(eqs, cols), rows, cols, 1, algo)
        piv = set(int(c) for c in res["pivcols"])
        free = [c for c in range(cols) if c not in piv]
        assert all(not (sp.origin >> f) & 1 for f in free)
        own = []
        for b in sp.basis:
            fs = [f for f in free if (b >> f) & 1]
            assert len(fs) == 1
            own.append(fs[0])
        order = list(range(cols))
        for i, c in enumerate(res["pivcols"]):
            order[i], order[c] = order[c], order[i]
        assert own == order[len(piv):]


def test_plain_equals_m4rm_medium():
    rng = random.Random(7)
    shapes = [(70, 64, .5, None, 0), (129, 128, .5, 100, 0), (300, 200, .1, None, 20), (513, 512, .5, 300, 0),
              (700, 650, .02, None, 30), (1030, 1025, .5, None, 0)]
    for rows, cols, dens, cap, zr in shapes:
        for consistent in (True, False):
            eqs = random_system(rng, rows, cols, dens, cap, consistent, zr)
            aug = O.eqs_to_aug(eqs, cols)
            a = O.solve_words(aug, rows, cols, 1, 0)
            b = O.solve_words(aug, rows, cols, 1, 1)
            assert a["status"] == b["status"] and a["rank"] == b["rank"]
            assert np.array_equal(a["pivcols"], b["pivcols"])
            if a["status"] == 0:
                assert np.array_equal(a["origin"], b["origin"]) and np.array_equal(a["basis"], b["basis"])
                assert O.check_solution(aug, rows, cols, a["origin"]) == 0


def test_sign_and_high_bits_ignored():
    # _internal.c:5-16 reads digit magnitudes only; bits above `cols` are never visited (:414)
    eqs = [15, 20, 11, 0]
    noisy = [-15, 20 | (1 << 5), 11 | (1 << 70), 0]
    a, b = O.m4ri_solve(eqs, 4, 1), O.m4ri_solve(noisy, 4, 1)
    assert (a.origin, a.basis) == (b.origin, b.basis)


def test_argument_errors():
    with pytest.raises(TypeError):
        O.m4ri_solve((1, 2), 2, 0)
    with pytest.raises(ValueError):
        O.m4ri_solve([1, 2], 0, 0)
    with pytest.raises(ValueError):
        O.m4ri_solve([1, 2], 2, 2)
    with pytest.raises(ValueError):
        O.m4ri_solve([1], 2, 0)
    with pytest.raises(TypeError):
        O.m4ri_solve([1, "x"], 2, 0)


def test_synthetic_generator_spec():
    """The C generator follows the written spec: word = mix64(mix64(seed) ^ (r<<20 | w))."""
    M = (1 << 64) - 1

    def mix64(x):
        x = (x + 0x9E3779B97F4A7C15) & M
        x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & M
        x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & M
        return x ^ (x >> 31)
    rows, cols, seed0 = 37, 200, 99
    aug = O.gen_synthetic(rows, cols, seed0)
    seed = mix64(seed0)
    cw = (cols + 63) // 64
    last = (1 << (cols % 64)) - 1
    xs = [mix64(seed ^ ((0xFFFFF << 20) | w)) for w in range(cw)]
    xs[-1] &= last
    assert [int(v) for v in O.planted_solution(cols, seed0)] == xs
    for r in range(rows):
        par = 0
        for w in range(cw):
            a = mix64(seed ^ ((r << 20) | w))
            if w == cw - 1:
                a &= last
            got = int(aug[r, w]) & (last if w == cw - 1 else M)
            assert got == a
            par ^= a & xs[w]
        assert (int(aug[r, cols // 64]) >> (cols % 64)) & 1 == bin(par).count("1") % 2
    assert O.check_solution(aug, rows, cols, O.planted_solution(cols, seed0)) == 0


def test_enumeration_orders():
    # Gray order for dim <= 64 (_internal.c:101-122), binary counter above (_internal.c:63-91)
    basis = tuple(1 << (i + 1) for i in range(3))
    sp = O.OracleSpace(1, basis)
    assert list(sp) == [1 ^ sum(basis[i] for i in range(3) if ((g ^ (g >> 1)) >> i) & 1) for g in range(8)]
    big = O.OracleSpace(0, tuple(1 << i for i in range(66)))
    it = iter(big)
    assert [next(it) for _ in range(5)] == [0, 1, 2, 3, 4]
