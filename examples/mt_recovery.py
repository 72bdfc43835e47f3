"""MT19937 state recovery from random.getrandbits outputs (BASELINE configs[2]; mirrors the
scenario of the reference's examples/mt.py with gf2bv_amd's own API and models)."""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gf2bv_amd import LinearSystem, PackedLinearSystem
from tests.harness_models import MT19937


def recover(bs, samples=None, system=LinearSystem):
    rand = random.Random(3142)
    state = tuple(rand.getstate()[1][:-1])
    eff = ((bs - 1) & bs) or bs
    samples = 624 * 32 // eff if samples is None else samples
    out = [rand.getrandbits(bs) for _ in range(samples)]
    lin = system([32] * 624)
    mt = lin.gens()
    rng = MT19937(mt)
    t0 = time.perf_counter()
    zeros = [rng.getrandbits(bs) ^ o for o in out] + [mt[0] ^ 0x80000000]
    t1 = time.perf_counter()
    sol = lin.solve_one(zeros)
    t2 = time.perf_counter()
    assert sol == state
    check = MT19937(sol)
    assert all(check.getrandbits(bs) == o for o in out)
    print(f"{system.__name__:18s} bs={bs:5d} generate {t1 - t0:6.2f}s  solve_one {t2 - t1:6.3f}s  ok")


if __name__ == "__main__":
    for bs, samples in ((32, None), (17, None), (9, None), (1, None), (1337, 19968 // 1337 + 10), (137, 19968 // 137 + 60)):
        recover(bs, samples)
        recover(bs, samples, PackedLinearSystem)       # the same scenario on bit matrices
