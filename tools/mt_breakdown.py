"""MT19937 recovery (bs = 32), packed rows -> gf2bv_solve_digits directly: where the solve_one wall time goes."""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gf2bv_amd import PackedLinearSystem, hip
from tests.harness_models import MT19937
rand = random.Random(3142)
out = [rand.getrandbits(32) for _ in range(624)]
pk = PackedLinearSystem([32] * 624)
mt = pk.gens()
rng = MT19937(mt)
zeros = [rng.getrandbits(32) ^ o for o in out] + [mt[0] ^ 0x80000000]
for rep in range(3):
    t0 = time.perf_counter()
    rows = np.ascontiguousarray(pk.get_rows(zeros))
    t1 = time.perf_counter()
    n, w = rows.shape
    off = np.arange(n + 1, dtype=np.int64) * (2 * w)
    sol = hip.solve_digits(rows.view(np.uint32).reshape(-1), off, 32, n, 19968, 0)
    t2 = time.perf_counter()
    s = sol.stats
    print(f"get_rows {1e3*(t1-t0):.1f} ms | solve_digits {1e3*(t2-t1):.1f} ms: pack(H2D+kernel) {s['ms_pack']:.1f} eliminate {s['ms_eliminate']:.1f} "
          f"backsub {s['ms_backsub']:.1f} export {s['ms_export']:.1f} total_host {s['ms_total']:.1f} fast_blocks {s['fast_blocks']}", flush=True)
