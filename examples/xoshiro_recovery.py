"""xoshiro256** seed recovery with solve_all (BASELINE configs[4])."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gf2bv_amd import LinearSystem
from tests.harness_models import Xoshiro256starstar

gen = Xoshiro256starstar.generate()
secret = tuple(gen.s)
outs = [gen() for _ in range(10)]
lin = LinearSystem([64] * 4)
sym = Xoshiro256starstar(lin.gens())
zeros = [sym.step() ^ Xoshiro256starstar.untemper(o) for o in outs]
t0 = time.perf_counter()
sols = list(lin.solve_all(zeros))
print(f"solve_all: {len(sols)} solution(s) in {time.perf_counter() - t0:.4f}s")
assert sols == [secret]
check = Xoshiro256starstar(list(sols[0]))
assert [check() for _ in range(10)] == outs
print("recovered", [hex(v) for v in sols[0]])
