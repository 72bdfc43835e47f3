"""MT19937 state recovery (examples/mt.py, bs = 32): where the wall time goes with the tuple-of-int front-end
(LinearSystem) and with the packed one (PackedLinearSystem)."""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gf2bv_amd import LinearSystem, PackedLinearSystem
from tests.harness_models import MT19937
bs = 32
rand = random.Random(3142)
state = tuple(rand.getstate()[1][:-1])
out = [rand.getrandbits(bs) for _ in range(624)]
for name, cls in (("tuple-of-int", LinearSystem), ("tuple-of-int", LinearSystem), ("packed", PackedLinearSystem), ("packed", PackedLinearSystem)):      # (the first line pays the HIP initialisation)
    t0 = time.perf_counter()
    lin = cls([32] * 624)
    mt = lin.gens()
    rng = MT19937(mt)
    zeros = [rng.getrandbits(bs) ^ o for o in out] + [mt[0] ^ 0x80000000]
    t1 = time.perf_counter()
    sol = lin.solve_one(zeros)
    t2 = time.perf_counter()
    assert sol == state
    print(f"{name:13s}: zeros {1e3 * (t1 - t0):8.1f} ms   solve_one (equations -> device -> solution) {1e3 * (t2 - t1):7.1f} ms   "
          f"total {1e3 * (t2 - t0):8.1f} ms", flush=True)
