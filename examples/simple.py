"""128 unknowns, 126 equations (rank 125): solve_all / solve_one / evaluate (BASELINE configs[0])."""
import os, secrets, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gf2bv_amd import LinearSystem


def magic(x, y):
    m = (1 << 64) - 1
    return ((x ^ (y >> 22) ^ (x << 13)) & m) >> 3, ((y ^ (x >> 7) ^ (y << 5)) & m) >> 3, (x ^ y) & 0b101101


inp = secrets.randbits(64), secrets.randbits(64)
target = magic(*inp)
lin = LinearSystem((64, 64))
xs, ys = lin.gens()
zeros = [s ^ t for s, t in zip(magic(xs, ys), target)]
sols = list(lin.solve_all(zeros))
assert inp in sols and all(magic(*s) == target for s in sols)
one = lin.solve_one(zeros)
assert all(lin.evaluate(z, one) == 0 for z in zeros)
print(f"{len(sols)} solutions, input among them; solve_one -> {tuple(hex(v) for v in one)}")
