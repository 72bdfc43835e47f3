"""One MT19937 state-recovery solve (BASELINE configs[2]) for rocprofv3.  usage: profile_mt.py [bs]"""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gf2bv_amd import LinearSystem
from tests.harness_models import MT19937
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 32
rand = random.Random(3142)
state = tuple(rand.getstate()[1][:-1])
eff = ((bs - 1) & bs) or bs
out = [rand.getrandbits(bs) for _ in range(624 * 32 // eff)]
lin = LinearSystem([32] * 624)
mt = lin.gens()
rng = MT19937(mt)
zeros = [rng.getrandbits(bs) ^ o for o in out] + [mt[0] ^ 0x80000000]
for rep in range(2):
    t = time.perf_counter()
    sol = lin.solve_one(zeros)
    print(f"solve_one {1e3 * (time.perf_counter() - t):.1f} ms", "ok" if sol == state else "WRONG", flush=True)
