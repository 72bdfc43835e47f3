"""Device-side AffineSpace enumeration (gf2bv/_internal.c:101-122 Gray walk, :63-91 binary walk, :242-273
get) and kernel bases of more than 7 vectors (the blocked parity back-substitution in groups of right-hand sides):
order and values identical to the oracle's, bit for bit."""
import itertools
import random

import numpy as np
import pytest

from gf2bv_amd import LinearSystem, hip, m4ri_solve
from oracle import gf2_oracle as O
from tests.systems import random_system

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert hip.device_count() >= 1, "gpu tests need an MI355X"


def _space(rng, dim, words):
    origin = np.array([rng.getrandbits(64) for _ in range(words)], dtype=np.uint64)
    basis = np.array([[rng.getrandbits(64) for _ in range(words)] for _ in range(dim)], dtype=np.uint64).reshape(dim, words)
    return origin, basis


@pytest.mark.parametrize("dim,words", [(0, 1), (1, 1), (5, 3), (12, 17), (16, 40), (64, 2), (70, 3)])
def test_enumerate_matches_oracle_walks(dim, words):
    rng = random.Random(dim * 131 + words)
    origin, basis = _space(rng, dim, words)
    sp = O.OracleSpace(O.words_to_int(origin), tuple(O.words_to_int(b) for b in basis))
    total = 1 << min(dim, 20)
    n = min(total, 3000)
    # Gray order = the oracle's iteration order (dimension <= 64); binary order = get(n)
    if dim <= 64:
        want = [O.int_to_words(v, words) for v in itertools.islice(iter(sp), n)]
        got = hip.space_enumerate(origin, basis, 0, n, gray=True)
        assert np.array_equal(got, np.stack(want))
    first = 0 if total <= n else rng.randrange(total - n)
    got = hip.space_enumerate(origin, basis, first, n, gray=False)
    assert np.array_equal(got, np.stack([O.int_to_words(sp.get(first + k), words) for k in range(n)]))
    if dim >= 64:                                      # the last elements before the 64-bit index wraps
        got = hip.space_enumerate(origin, basis, (1 << 64) - 5, 5, gray=(dim == 64))
        idx = [(1 << 64) - 5 + k for k in range(5)]
        code = [(g ^ (g >> 1)) if dim == 64 else g for g in idx]
        assert np.array_equal(got, np.stack([O.int_to_words(sp.get(c), words) for c in code]))


@pytest.mark.parametrize("rows,cols,cap", [(900, 700, 688), (2200, 2100, 2080), (1500, 1300, 1244), (1500, 1300, 1225)])
def test_kernel_bases_beyond_seven_vectors(rows, cols, cap):
    """dim = cols - cap = 12, 20, 56 (parity path in groups of 8 right-hand sides) and 75 (table sweeps over Y)."""
    rng = random.Random(rows + cap)
    eqs = random_system(rng, rows, cols, .5, cap, True, 0)
    aug = O.eqs_to_aug(eqs, cols)
    want = O.solve_words(aug, rows, cols, 1)
    got = hip.solve_words(aug, rows, cols, 1)
    assert got.status == want["status"] == 0 and got.rank == want["rank"] and got.dimension == cols - want["rank"]
    assert got.dimension >= cols - cap
    assert np.array_equal(got.origin, want["origin"]) and np.array_equal(got.basis, want["basis"])


def test_solve_all_order_through_the_device_chunks():
    """LinearSystem.solve_all on a rank-deficient system whose solution space is enumerated from device-filled chunks
    (dimension 12, 41 words per element -> one chunk of 4096 elements), against the oracle's iteration order."""
    rng = random.Random(2024)
    sizes = [64] * 40 + [37]
    cols = sum(sizes)
    lin = LinearSystem(sizes)
    eqs = random_system(rng, cols + 30, cols, .5, cols - 12, True, 0)
    space = m4ri_solve(eqs, cols, 1)
    ref = O.m4ri_solve(eqs, cols, 1)
    assert space.dimension == ref.dimension >= 12
    n = 1 << ref.dimension if ref.dimension <= 13 else 9000
    got = list(itertools.islice(iter(space), n))
    assert got == list(itertools.islice(iter(ref), n))
    assert [lin.convert_sol(s) for s in got[:5]] == [lin.convert_sol(s) for s in itertools.islice(iter(ref), 5)]
    if ref.dimension <= 13:
        assert sum(1 for _ in space) == 1 << ref.dimension          # a second iterator ends where it should
    # small spaces (README 4-variable system) keep the host walk: same contract
    sp = m4ri_solve([15, 20, 11, 0], 4, 1)
    assert list(sp) == [0b0001, 0b0100] and sp.get(1) == 0b0100
