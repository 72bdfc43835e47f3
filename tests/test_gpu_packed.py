"""Packed symbolic front-end on the GPU path: PackedLinearSystem -> _internal.m4ri_solve_packed -> gf2bv_solve_digits
(32-bit digits, device pack kernel).  Known answers of the reference's examples (examples/mt.py:38: the recovered
state equals random.Random(3142)'s; examples/xoshiro.py:16) and equality with the list-of-int path."""
import random

import pytest

from gf2bv_amd import LinearSystem, PackedLinearSystem, hip
from tests.harness_models import MT19937, Xoshiro256starstar
from tests.systems import random_system

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert hip.device_count() >= 1, "gpu tests need an MI355X"


@pytest.mark.parametrize("bs,samples", [(32, None), (17, None), (137, 19968 // 137 + 60)])
def test_mt19937_recovery_through_the_packed_front_end(bs, samples):
    rand = random.Random(3142)
    state = tuple(rand.getstate()[1][:-1])
    eff = ((bs - 1) & bs) or bs
    samples = 624 * 32 // eff if samples is None else samples
    out = [rand.getrandbits(bs) for _ in range(samples)]
    pk = PackedLinearSystem([32] * 624)
    mt = pk.gens()
    rng = MT19937(mt)
    zeros = [rng.getrandbits(bs) ^ o for o in out] + [mt[0] ^ 0x80000000]
    sol = pk.solve_one(zeros)
    assert sol == state                                    # examples/mt.py:38
    assert all(pk.evaluate(z, sol) == 0 for z in zeros[:8])
    follow = MT19937(sol).to_python_random()
    assert [follow.getrandbits(bs) for _ in range(samples)] == out


def test_packed_and_int_paths_agree_on_rank_deficient_systems():
    rng = random.Random(12)
    sizes = [40, 24, 64, 13]
    cols = sum(sizes)
    lin, pk = LinearSystem(sizes), PackedLinearSystem(sizes)
    eqs = random_system(rng, cols + 20, cols, .5, cols - 5, True, 0)
    want = list(lin.solve_all(eqs))
    got = list(pk.solve_all(eqs))                           # bare equation ints go in as rows
    assert got == want and len(got) >= 32
    assert pk.solve_one(eqs) == lin.solve_one(eqs) == want[0]
    bad = list(eqs)
    bad[3] ^= 1
    assert pk.solve_one(bad) == lin.solve_one(bad)
    sp, st = pk.solve_raw_space(eqs), lin.solve_raw_space(eqs)
    assert (sp.dimension, sp.origin, sp.basis) == (st.dimension, st.origin, st.basis)
    with pytest.raises(Exception) as ei:
        next(pk.solve_all(eqs, max_dimension=2))
    assert type(ei.value).__name__ == "DimensionTooLargeError" and ei.value.space.dimension == sp.dimension


def test_xoshiro_recovery_packed():
    r = random.Random(1)
    state = [r.getrandbits(64) for _ in range(4)]
    gen = Xoshiro256starstar(list(state))
    outs = [gen() for _ in range(10)]
    pk = PackedLinearSystem([64] * 4)
    sym = Xoshiro256starstar(pk.gens())
    zeros = [sym.step() ^ Xoshiro256starstar.untemper(o) for o in outs]
    assert list(pk.solve_all(zeros)) == [tuple(state)]
