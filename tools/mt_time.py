import os, random, sys, time
sys.path.insert(0, "/root/repo")
from gf2bv_amd import LinearSystem, _internal
from tests.harness_models import MT19937
bs = 32
rand = random.Random(3142)
state = tuple(rand.getstate()[1][:-1])
out = [rand.getrandbits(bs) for _ in range(624)]
lin = LinearSystem([32] * 624)
mt = lin.gens()
rng = MT19937(mt)
zeros = [rng.getrandbits(bs) ^ o for o in out] + [mt[0] ^ 0x80000000]
for rep in range(3):
    t0 = time.perf_counter()
    eqs = lin.get_eqs(zeros)
    t1 = time.perf_counter()
    eqs2 = eqs + [0] * max(0, lin._cols - len(eqs)) if hasattr(lin, "_cols") else eqs
    r = _internal.m4ri_solve(eqs2, 19968, 0)
    t2 = time.perf_counter()
    sol = lin.solve_one(zeros)
    t3 = time.perf_counter()
    print(f"get_eqs {1e3*(t1-t0):.1f} ms ({len(eqs)} eqs), m4ri_solve {1e3*(t2-t1):.1f} ms, whole solve_one {1e3*(t3-t2):.1f} ms", flush=True)
