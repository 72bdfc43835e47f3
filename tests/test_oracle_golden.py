"""Pins the oracle (and the harness' own system builders) against the reference's known answers:
golden equation lists captured from the reference's Python layer (tests/golden/make_golden.py)
and the KATs the reference's example scripts assert.  No GPU."""
import random

import pytest

from oracle import gf2_oracle as O
from tests import harness as H

G = H.GOLDEN


def _eqs(entry):
    return [int(e, 16) for e in entry["eqs"]]


def test_readme4_golden():
    e = G["readme4"]
    assert H.fingerprint(_eqs(e), 4) == e["sha256"]
    sp = O.m4ri_solve(_eqs(e), 4, 1)
    assert sp.origin == e["expect"]["origin"] and list(sp.basis) == e["expect"]["basis"]
    lin = H.LinearSystem(e["sizes"])
    assert [list(lin.convert_sol(s)) for s in sp] == e["expect"]["solve_all"]
    assert list(lin.convert_sol(O.m4ri_solve(_eqs(e), 4, 0))) == e["expect"]["solve_one"]


def test_simple_golden():
    # examples/simple.py:16-27 -- every solution satisfies magic(x,y) == expected; evaluate(z,sol) == 0
    for key, inp in (("simple_linear", None), ("simple_affine", True)):
        e = G[key]
        lin, zeros, expected = H.simple_system(tuple(int(v, 16) for v in e["input"]) if inp else None)
        eqs = H.padded_eqs(lin, zeros)
        assert eqs == _eqs(e) and H.fingerprint(eqs, 128) == e["sha256"]
        sp = O.m4ri_solve(eqs, 128, 1)
        sols = [lin.convert_sol(s) for s in sp]
        assert len(sols) == len(set(sols)) == 2 ** sp.dimension
        if key == "simple_linear":
            assert len(sols) == e["expect"]["n_solutions"]
            assert all(q & 1 == 0 for q in eqs)                  # homogeneous (examples/simple.py:35)
        for sol in sols:
            assert H.magic(*sol) == tuple(expected)
        one = lin.convert_sol(O.m4ri_solve(eqs, 128, 0))
        assert H.magic(*one) == tuple(expected)
        assert all(lin.evaluate(z, one) == 0 for z in zeros)


def test_xoshiro_golden():
    e = G["xoshiro"]
    lin, zeros, state, outs = H.xoshiro_system(1, 10)
    eqs = H.padded_eqs(lin, zeros)
    assert eqs == _eqs(e) and H.fingerprint(eqs, 256) == e["sha256"]
    assert [hex(o) for o in outs] == e["outputs"]
    sp = O.m4ri_solve(eqs, 256, 1)
    sols = [lin.convert_sol(s) for s in sp]
    assert sols == [tuple(int(v, 16) for v in e["expect"]["solve_all"][0])] == [state]
    gen = H.Xoshiro256starstar(list(sols[0]))
    assert [gen() for _ in range(10)] == outs                    # examples/xoshiro.py:16


@pytest.mark.parametrize("bs,samples", [H.MT_VARIANTS[0], H.MT_VARIANTS[3], H.MT_VARIANTS[4]])
def test_mt19937_kat(bs, samples):
    """examples/mt.py:21-22,38: sol == state of random.Random(3142) -- pins the oracle's result
    independently of any tie-breaking (full column rank -> unique solution)."""
    lin, zeros, state, out = H.mt19937_system(bs, samples)
    eqs = H.padded_eqs(lin, zeros)
    meta = G["mt19937"]["variants"][str(bs)]
    assert (len(eqs), lin._cols) == (meta["rows"], meta["cols"])
    assert H.fingerprint(eqs, lin._cols) == meta["sha256"]
    raw = O.m4ri_solve(eqs, lin._cols, 0)
    sol = lin.convert_sol(raw)
    assert sol == state
    rng = H.MT19937(sol)
    assert all(rng.getrandbits(bs) == o for o in out[:50])       # examples/mt.py:42-45
    py = H.MT19937(sol).to_python_random()
    assert all(py.getrandbits(bs) == o for o in out[:50])
    assert random.Random(3142).getrandbits(bs) == out[0]


def test_quadratic_golden():
    """QuadraticSystem (gf2bv/__init__.py:290-408): this repo's restatement builds the reference's equation list bit for
    bit; the oracle's solution space, filtered by convert_sol, is the brute-force solution set of the quadratic system."""
    e = G["quadratic"]
    q, zeros = H.quadratic_small_system(e["consts"])
    eqs = H.padded_eqs(q, zeros)
    assert [hex(v) for v in eqs] == e["eqs"] and H.fingerprint(eqs, q._cols) == e["sha256"]
    assert q._cols == e["cols"] == 5 + 10 and len(q.gens()) == 2
    sp = O.m4ri_solve(eqs, q._cols, 1)
    sols = sorted(list(s) for s in (q.convert_sol(raw) for raw in sp) if s is not None)
    assert sols == e["expect"]["solutions"] and e["planted"] in sols
    x, y = q.gens()
    for sol in sols:                    # evaluate() works on the linear part (as in the reference): the asserted linear bit
        assert q.evaluate(x[1] ^ y[0], tuple(sol)) == e["consts"][3]


def test_quadratic_system_semantics():
    q = H.QuadraticSystem([3, 2])
    x, y = q.gens()
    n = 5
    rng = random.Random(9)

    def value(eq, assign):              # equation int under an assignment of the 5 bits, products filled in
        quad, pos = 0, 0
        for i in range(n):
            for j in range(i):
                quad |= (((assign >> i) & 1) & ((assign >> j) & 1)) << pos
                pos += 1
        return bin(eq & ((((quad << n) | assign) << 1) | 1)).count("1") & 1

    for _ in range(100):                # products of linear forms without constant terms
        a, b = rng.getrandbits(n) << 1, rng.getrandbits(n) << 1
        eq = q._mul_bit(a, b)
        for assign in range(32):
            assert value(eq, assign) == (bin((a >> 1) & assign).count("1") & bin((b >> 1) & assign).count("1") & 1)
    with pytest.raises(ValueError):
        q.mul_bit(x, y[0])
    with pytest.raises(ValueError):
        q.bit_assert(x, 1)
    zs = q.bit_assert(x[0] ^ y[1], 1)
    assert len(zs) == n + 1 and zs[0] == (x[0] ^ y[1] ^ 1)._bits[0]
    # convert_sol: consistent product block -> the variables; inconsistent -> None
    lin = 0b10110
    quad, pos = 0, 0
    for i in range(n):
        for j in range(i):
            quad |= (((lin >> i) & 1) & ((lin >> j) & 1)) << pos
            pos += 1
    assert q.convert_sol(lin | (quad << n)) == (0b110, 0b10)
    assert q.convert_sol(lin | ((quad ^ 1) << n)) is None
    import pickle
    q2 = pickle.loads(pickle.dumps(q))
    assert isinstance(q2, H.QuadraticSystem) and q2._quad_sizes == [3, 2] and q2._cols == 15


@pytest.mark.parametrize("name", sorted(H.NLFSR_KINDS))
def test_nlfsr_equations_golden(name):
    """examples/nlfsr.py: the linearised annihilator equations of the first 3000 outputs, as the reference's Python
    layer builds them (fingerprint)."""
    v = G["nlfsr"]["variants"][name]
    q, zeros, _ = H.nlfsr_system(name, v["outputs"])
    eqs = q.get_eqs(zeros)
    assert (len(eqs), q._cols) == (v["equations"], v["cols"])
    assert H.fingerprint(eqs, q._cols) == v["sha256"]
