#!/usr/bin/env python3
"""Regenerates tests/golden/*.json.  Runs ONLY in the build container (needs /root/reference).

The reference's Python layer (gf2bv/__init__.py, gf2bv/crypto/*.py) is imported from
/root/reference with this repo's own `_internal` extension standing in for `gf2bv._internal`
(the reference's C file cannot be built: no M4RI).  Only the *input side* of the hot path is
exercised that way: zeros -> get_eqs -> the padded equation list handed to m4ri_solve.  The
expected *outputs* are the reference's own known answers (examples/mt.py:21-22,38:
sol == Random(3142) state; examples/xoshiro.py:16: the generating state; README.md:32-44 /
SURVEY 8a-S hand-derived KAT).  Nothing of the reference's source text is stored -- only
equation integers (data) and sha256 fingerprints.
"""
import hashlib
import importlib
import json
import os
import random
import shutil
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"


def reference_package():
    tmp = tempfile.mkdtemp(prefix="gf2bv_ref_")
    pkg = os.path.join(tmp, "gf2bv")
    os.makedirs(os.path.join(pkg, "crypto"))
    for rel in ("__init__.py", "crypto/__init__.py", "crypto/mt.py", "crypto/xoshiro.py", "crypto/lfsr.py"):
        os.symlink(os.path.join(REF, "gf2bv", rel), os.path.join(pkg, rel))
    import sysconfig
    ext = "_internal" + sysconfig.get_config_var("EXT_SUFFIX")
    shutil.copy(os.path.join(ROOT, "gf2bv_amd", ext), os.path.join(pkg, ext))
    shutil.copy(os.path.join(ROOT, "gf2bv_amd", "libgf2bv_hip.so"), os.path.join(pkg, "libgf2bv_hip.so"))
    sys.path.insert(0, tmp)
    return importlib.import_module("gf2bv"), tmp


def fingerprint(eqs, cols):
    h = hashlib.sha256()
    nb = (cols + 1 + 7) // 8
    for e in eqs:
        h.update(e.to_bytes(nb, "little"))
    return h.hexdigest()


def padded(lin, zeros):
    eqs = lin.get_eqs(zeros)
    if lin._cols > len(eqs):
        eqs += [0] * (lin._cols - len(eqs))
    return eqs


def main():
    gf2bv, tmp = reference_package()
    from gf2bv.crypto.mt import MT19937
    from gf2bv.crypto.xoshiro import Xoshiro256starstar
    out = {}

    # README.md:32-44 -- 4 unknowns, 3 equations
    lin = gf2bv.LinearSystem([1, 1, 1, 1])
    a, b, c, d = lin.gens()
    zeros = [a ^ b ^ c ^ 1, b ^ d, a ^ c ^ 1]
    eqs = padded(lin, zeros)
    out["readme4"] = {"sizes": [1, 1, 1, 1], "cols": 4, "eqs": [hex(e) for e in eqs],
                      "sha256": fingerprint(eqs, 4),
                      "expect": {"origin": 0b0001, "basis": [0b0101], "solve_all": [[1, 0, 0, 0], [0, 0, 1, 0]],
                                 "solve_one": [1, 0, 0, 0]},
                      "source": "README.md:32-44; expected values hand-derived in SURVEY.md 8a-S"}

    # examples/simple.py simple_linear -- 128 unknowns, 126 equations, rank 125
    def magic(x, y):
        m = (1 << 64) - 1
        return ((x ^ (y >> 22) ^ (x << 13)) & m) >> 3, ((y ^ (x >> 7) ^ (y << 5)) & m) >> 3, (x ^ y) & 0b101101
    lin = gf2bv.LinearSystem((64, 64))
    xs, ys = lin.gens()
    eqs = padded(lin, list(magic(xs, ys)))
    out["simple_linear"] = {"sizes": [64, 64], "cols": 128, "eqs": [hex(e) for e in eqs],
                            "sha256": fingerprint(eqs, 128),
                            "expect": {"n_solutions": 8, "magic": [0, 0, 0]},
                            "source": "examples/simple.py:29-36 (every solution must satisfy magic(x,y)==(0,0,0))"}
    # examples/simple.py simple_affine with a fixed input instead of secrets.randbits
    inp = (random.Random(7).getrandbits(64), random.Random(8).getrandbits(64))
    z = magic(*inp)
    eqs = padded(lin, [s ^ v for s, v in zip(magic(xs, ys), z)])
    out["simple_affine"] = {"sizes": [64, 64], "cols": 128, "eqs": [hex(e) for e in eqs],
                            "sha256": fingerprint(eqs, 128), "input": [hex(v) for v in inp],
                            "expect": {"magic": list(z)},
                            "source": "examples/simple.py:39-49 with inp fixed to Random(7)/Random(8).getrandbits(64)"}

    # examples/xoshiro.py with the state fixed to 4 x Random(1).getrandbits(64)
    r = random.Random(1)
    state = [r.getrandbits(64) for _ in range(4)]
    xos = Xoshiro256starstar(list(state))
    outs = [xos() for _ in range(10)]
    lin = gf2bv.LinearSystem([64] * 4)
    xs2 = Xoshiro256starstar(lin.gens())
    eqs = padded(lin, [xs2.step() ^ Xoshiro256starstar.untemper(o) for o in outs])
    out["xoshiro"] = {"sizes": [64] * 4, "cols": 256, "eqs": [hex(e) for e in eqs], "sha256": fingerprint(eqs, 256),
                      "outputs": [hex(o) for o in outs],
                      "expect": {"solve_all": [[hex(s) for s in state]]},
                      "source": "examples/xoshiro.py:6-16, state = 4 x random.Random(1).getrandbits(64)"}

    # examples/mt.py mt19937(bs): only the fingerprint of the padded equation list is stored (50 MB of ints)
    mts = {}
    for bs, samples in ((32, None), (17, None), (9, None), (1, None), (1337, 19968 // 1337 + 10), (137, 19968 // 137 + 60)):
        rand = random.Random(3142)
        eff = ((bs - 1) & bs) or bs
        ns = 624 * 32 // eff if samples is None else samples
        o = [rand.getrandbits(bs) for _ in range(ns)]
        lin = gf2bv.LinearSystem([32] * 624)
        mt = lin.gens()
        rng = MT19937(mt)
        eqs = padded(lin, [rng.getrandbits(bs) ^ v for v in o] + [mt[0] ^ 0x80000000])
        mts[str(bs)] = {"samples": ns, "rows": len(eqs), "cols": lin._cols, "sha256": fingerprint(eqs, lin._cols)}
    out["mt19937"] = {"seed": 3142, "variants": mts,
                      "expect": "solve_one(zeros) == tuple(random.Random(3142).getstate()[1][:-1])",
                      "source": "examples/mt.py:19-54"}

    # QuadraticSystem (gf2bv/__init__.py:290-408): two variables of 3 + 2 bits -> 5 linear + 10 product unknowns.
    # Expected solutions: brute force over the 32 assignments of the ORIGINAL quadratic equations (no solver involved).
    q = gf2bv.QuadraticSystem([3, 2])
    x, y = q.gens()
    planted = (0b101, 0b10)
    def val(bv):
        return q.evaluate(bv, planted)
    polys = [
        lambda X, Y: (X[0] & Y[1]) ^ X[2],
        lambda X, Y: ((X[0] ^ X[1]) & (Y[0] ^ X[2])) ^ Y[1],
        lambda X, Y: (X[1] & X[2]) ^ (Y[0] & Y[1]) ^ X[0],
        lambda X, Y: X[1] ^ Y[0],
    ]
    def bits(v, n):
        return [(v >> i) & 1 for i in range(n)]
    consts = [p_(bits(planted[0], 3), bits(planted[1], 2)) for p_ in polys]
    zeros = [q.mul_bit(x[0], y[1]) ^ x[2] ^ consts[0],
             q.mul_bit(x[0] ^ x[1], y[0] ^ x[2]) ^ y[1] ^ consts[1],
             q.mul_bit(x[1], x[2]) ^ q.mul_bit(y[0], y[1]) ^ x[0] ^ consts[2]]
    zeros += list(q.bit_assert(x[1] ^ y[0], consts[3]))
    eqs = padded(q, zeros)
    sols = sorted([xv, yv] for xv in range(8) for yv in range(4)
                  if [p_(bits(xv, 3), bits(yv, 2)) for p_ in polys] == consts)
    out["quadratic"] = {"sizes": [3, 2], "cols": q._cols, "eqs": [hex(e) for e in eqs], "sha256": fingerprint(eqs, q._cols),
                        "planted": list(planted), "consts": consts,
                        "expect": {"solutions": sols},
                        "source": "gf2bv/__init__.py:290-408 (mul_bit, bit_assert); solutions by brute force over the 32 assignments"}

    # examples/nlfsr.py: the linearised equations of the filtered LFSR (first 3000 outputs; fingerprint only)
    from gf2bv.crypto.lfsr import FibonacciLFSR, GaloisLFSR
    n_bits, taps, select = 128, 0xD670201BAC7515352A273372B2A95B23, (13, 24, 35, 46, 57)
    def filt(x0, x1, x2, x3, x4):
        return (x0 & x1) ^ (x0 & x1 & x3 & x4) ^ x0 ^ x1 ^ x2
    nl = {}
    for name, kind, seed in (("galois", GaloisLFSR, 1), ("fibonacci", FibonacciLFSR, 2)):
        secret = random.Random(seed).getrandbits(n_bits)
        lfsr = kind(n_bits, taps, secret)
        stream = []
        for _ in range(3000):
            lfsr()
            stream.append(filt(*[(lfsr.state >> i) & 1 for i in select]))
        qs = gf2bv.QuadraticSystem([n_bits])
        (xv,) = qs.gens()
        sym = kind(n_bits, taps, xv)
        zs = []
        for bit in stream:
            sym()
            if bit:
                x0, x1, x2, _, _ = [sym.state[i] for i in select]
                zs.append(qs.mul_bit(x0, x1) ^ x0 ^ qs.mul_bit(x1, x2) ^ x1 ^ x2 ^ 1)
        e = qs.get_eqs(zs)
        nl[name] = {"seed": seed, "outputs": 3000, "equations": len(e), "cols": qs._cols, "sha256": fingerprint(e, qs._cols)}
    out["nlfsr"] = {"variants": nl, "taps": hex(taps), "select": list(select),
                    "expect": "solve_all(zeros of 2**14 + 1000 outputs) == [(Random(seed).getrandbits(128),)]",
                    "source": "examples/nlfsr.py:9-64 with the secret fixed to Random(seed).getrandbits(128)"}

    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(out, f, indent=1)
    shutil.rmtree(tmp, ignore_errors=True)
    print("wrote golden.json:", {k: (v.get("sha256") or "...")[:16] for k, v in out.items()})


if __name__ == "__main__":
    main()
