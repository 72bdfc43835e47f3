"""Round 5: several MT19937 recovery systems of one shape as ONE m4ri_solve_many call (a lock-step gang: general panel steps) against
the same systems through m4ri_solve one after the other (k_block_sparse).  usage: mt_many_time.py [nsys] [bs]"""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gf2bv_amd import LinearSystem, _internal
from tests.harness_models import MT19937
nsys = int(sys.argv[1]) if len(sys.argv) > 1 else 8
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 32
systems, states = [], []
for s in range(nsys):
    rand = random.Random(3142 + s)
    states.append(tuple(rand.getstate()[1][:-1]))
    eff = ((bs - 1) & bs) or bs  # as the reference's example counts the informative bits of an output
    obs = [rand.getrandbits(bs) for _ in range(624 * 32 // eff)]
    lin = LinearSystem([32] * 624)
    mt = lin.gens()
    sym = MT19937(mt)
    eqs = lin.get_eqs([sym.getrandbits(bs) ^ o for o in obs] + [mt[0] ^ 0x80000000])
    eqs += [0] * max(0, lin._cols - len(eqs))
    systems.append(eqs)
cols = lin._cols
for rep in range(3):
    t0 = time.perf_counter()
    many = _internal.m4ri_solve_many(systems, cols, 0)
    t1 = time.perf_counter()
    one = [_internal.m4ri_solve(e, cols, 0) for e in systems]
    t2 = time.perf_counter()
    ok = many == one and all(lin.convert_sol(r) == st for r, st in zip(one, states))
    thr = os.environ.get("GF2BV_BATCH_THREADS", "4") if os.environ.get("GF2BV_SPARSE_BATCH", "1") != "0" else "gangs"
    print(f"[{thr}] ", end="")
    print(f"{nsys} x MT19937 bs={bs}: m4ri_solve_many {1e3 * (t1 - t0):.1f} ms ({1e3 * (t1 - t0) / nsys:.1f} per system), {nsys} x m4ri_solve {1e3 * (t2 - t1):.1f} ms ({1e3 * (t2 - t1) / nsys:.1f} per system), equal and known answers: {ok}", flush=True)
