"""Builders for the reference's example systems, written against gf2bv_amd's own API
(mirrors examples/mt.py, examples/xoshiro.py, examples/simple.py of maple3142/gf2bv)."""
from __future__ import annotations

import hashlib
import json
import os
import random

from gf2bv_amd import LinearSystem, QuadraticSystem
from tests.harness_models import MT19937, FibonacciLFSR, GaloisLFSR, Xoshiro256starstar

GOLDEN = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden.json")))
MT_VARIANTS = ((32, None), (17, None), (9, None), (1, None), (1337, 19968 // 1337 + 10), (137, 19968 // 137 + 60))


def fingerprint(eqs, cols) -> str:
    h = hashlib.sha256()
    nb = (cols + 1 + 7) // 8
    for e in eqs:
        h.update(e.to_bytes(nb, "little"))
    return h.hexdigest()


def padded_eqs(lin: LinearSystem, zeros) -> list:
    eqs = lin.get_eqs(zeros)
    if lin._cols > len(eqs):
        eqs += [0] * (lin._cols - len(eqs))
    return eqs


def mt19937_system(bs: int, samples=None):
    """examples/mt.py:19-33: outputs of random.Random(3142), plus the mt[0] msb equation."""
    rand = random.Random(3142)
    state = tuple(rand.getstate()[1][:-1])
    eff = ((bs - 1) & bs) or bs
    samples = 624 * 32 // eff if samples is None else samples
    out = [rand.getrandbits(bs) for _ in range(samples)]
    lin = LinearSystem([32] * 624)
    mt = lin.gens()
    rng = MT19937(mt)
    zeros = [rng.getrandbits(bs) ^ o for o in out] + [mt[0] ^ 0x80000000]
    return lin, zeros, state, out


def xoshiro_system(seed: int = 1, n_out: int = 10):
    r = random.Random(seed)
    state = [r.getrandbits(64) for _ in range(4)]
    gen = Xoshiro256starstar(list(state))
    outs = [gen() for _ in range(n_out)]
    lin = LinearSystem([64] * 4)
    sym = Xoshiro256starstar(lin.gens())
    zeros = [sym.step() ^ Xoshiro256starstar.untemper(o) for o in outs]
    return lin, zeros, tuple(state), outs


def magic(x, y):
    m = (1 << 64) - 1
    return ((x ^ (y >> 22) ^ (x << 13)) & m) >> 3, ((y ^ (x >> 7) ^ (y << 5)) & m) >> 3, (x ^ y) & 0b101101


def simple_system(inp=None):
    lin = LinearSystem((64, 64))
    xs, ys = lin.gens()
    sym = magic(xs, ys)
    if inp is None:
        return lin, list(sym), (0, 0, 0)
    z = magic(*inp)
    return lin, [s ^ v for s, v in zip(sym, z)], z


# ---- QuadraticSystem (caller of the same path; gf2bv/__init__.py:290-408, examples/nlfsr.py) ----------------------
def quadratic_small_system(consts):
    """The 3 + 2-bit quadratic system of golden.json["quadratic"]: zeros for the given right-hand sides."""
    q = QuadraticSystem([3, 2])
    x, y = q.gens()
    zeros = [q.mul_bit(x[0], y[1]) ^ x[2] ^ consts[0],
             q.mul_bit(x[0] ^ x[1], y[0] ^ x[2]) ^ y[1] ^ consts[1],
             q.mul_bit(x[1], x[2]) ^ q.mul_bit(y[0], y[1]) ^ x[0] ^ consts[2]]
    zeros += list(q.bit_assert(x[1] ^ y[0], consts[3]))
    return q, zeros


NLFSR_BITS, NLFSR_TAPS, NLFSR_SELECT = 128, 0xD670201BAC7515352A273372B2A95B23, (13, 24, 35, 46, 57)
NLFSR_KINDS = {"galois": (GaloisLFSR, 1), "fibonacci": (FibonacciLFSR, 2)}


def nlfsr_filter(x0, x1, x2, x3, x4):
    return (x0 & x1) ^ (x0 & x1 & x3 & x4) ^ x0 ^ x1 ^ x2


def nlfsr_system(name: str, outputs: int):
    """examples/nlfsr.py with the secret fixed to Random(seed).getrandbits(128): one quadratic (annihilator) equation
    per output bit 1, linearised.  Returns (QuadraticSystem, zeros, secret)."""
    kind, seed = NLFSR_KINDS[name]
    secret = random.Random(seed).getrandbits(NLFSR_BITS)
    lfsr = kind(NLFSR_BITS, NLFSR_TAPS, secret)
    stream = []
    for _ in range(outputs):
        lfsr()
        stream.append(nlfsr_filter(*[(lfsr.state >> i) & 1 for i in NLFSR_SELECT]))
    q = QuadraticSystem([NLFSR_BITS])
    (x,) = q.gens()
    sym = kind(NLFSR_BITS, NLFSR_TAPS, x)
    zeros = []
    for bit in stream:
        sym()
        if bit:
            x0, x1, x2, _, _ = [sym.state[i] for i in NLFSR_SELECT]
            zeros.append(q.mul_bit(x0, x1) ^ x0 ^ q.mul_bit(x1, x2) ^ x1 ^ x2 ^ 1)
    return q, zeros, secret
