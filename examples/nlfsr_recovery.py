"""State recovery of a nonlinearly filtered 128-bit LFSR by linearisation (QuadraticSystem).

The filter f(x0..x4) = x0 x1 + x0 x1 x3 x4 + x0 + x1 + x2 has the degree-2 annihilator
g = x0 x1 + x0 + x1 x2 + x1 + x2 + 1 (f = 1 implies g = 0), so every output bit 1 gives one QUADRATIC equation in the
128 state bits; with the 8128 pairwise products as unknowns of their own that is a linear system of ~8700 equations
in 8256 unknowns -- one dense GF(2) solve on the GPU.  (Same experiment as the reference's examples/nlfsr.py.)
"""
import itertools, os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gf2bv_amd import QuadraticSystem
from tests.harness_models import FibonacciLFSR, GaloisLFSR

N_BITS, TAPS = 128, 0xD670201BAC7515352A273372B2A95B23
SELECT = (13, 24, 35, 46, 57)


def filter_bit(x0, x1, x2, x3, x4):
    return (x0 & x1) ^ (x0 & x1 & x3 & x4) ^ x0 ^ x1 ^ x2


def annihilator(x0, x1, x2, x3, x4):
    return (x0 & x1) ^ x0 ^ (x1 & x2) ^ x1 ^ x2 ^ 1


for bits in itertools.product((0, 1), repeat=5):
    assert not (filter_bit(*bits) and annihilator(*bits))


def keystream(lfsr, count):
    out = []
    for _ in range(count):
        lfsr()
        out.append(filter_bit(*[(lfsr.state >> i) & 1 for i in SELECT]))
    return out


def recover(kind, seed, count=2 ** 14 + 1000):
    secret = random.Random(seed).getrandbits(N_BITS)
    stream = keystream(kind(N_BITS, TAPS, secret), count)
    t0 = time.perf_counter()
    qsys = QuadraticSystem([N_BITS])
    (x,) = qsys.gens()
    sym = kind(N_BITS, TAPS, x)
    zeros = []
    for bit in stream:
        sym()
        if bit:
            x0, x1, x2, _, _ = [sym.state[i] for i in SELECT]
            zeros.append(qsys.mul_bit(x0, x1) ^ x0 ^ qsys.mul_bit(x1, x2) ^ x1 ^ x2 ^ 1)
    t1 = time.perf_counter()
    sols = list(qsys.solve_all(zeros))
    t2 = time.perf_counter()
    assert sols == [(secret,)], (kind.__name__, len(sols))
    assert qsys.solve_one(zeros) == (secret,)
    print(f"{kind.__name__:14s} {len(zeros)} equations x {qsys._cols} unknowns: generate {t1 - t0:.2f}s  solve_all {t2 - t1:.3f}s  ok")
    return secret


if __name__ == "__main__":
    recover(GaloisLFSR, 1)
    recover(FibonacciLFSR, 2)
