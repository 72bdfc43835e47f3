"""Packed symbolic front-end: PackedBitVec / PackedLinearSystem must produce, bit for bit, the equations
the tuple-of-int BitVec / LinearSystem produce (which tests/test_oracle_golden.py pins to the reference's own Python
layer), for every operator and for the PRNG models of the recovery examples."""
import random

import numpy as np
import pytest

from gf2bv_amd import BitVec, LinearSystem, PackedBitVec, PackedLinearSystem
from tests.harness_models import MT19937, GaloisLFSR, Xoshiro256starstar
from tests import harness as H


def _pair(sizes):
    return LinearSystem(sizes), PackedLinearSystem(sizes)


def _same(a: BitVec, p: PackedBitVec):
    assert len(a) == len(p) and a._bits == p._bits


def test_every_operator_matches_the_tuple_representation():
    rng = random.Random(3)
    lin, pk = _pair([32, 16, 32, 7])
    a, b, c, d = lin.gens()
    pa, pb, pc, pd = pk.gens()
    _same(a, pa)
    _same(d, pd)
    x, px = a ^ c ^ 0xDEADBEEF, pa ^ pc ^ 0xDEADBEEF
    _same(x, px)
    _same(0x1234 ^ b, 0x1234 ^ pb)
    for n in (0, 1, 5, 31, 32):
        _same(x >> n, px >> n)
        _same(x << n, px << n)
        _same(x.rotl(n), px.rotl(n))
        _same(x.rotr(n), px.rotr(n))
        _same(x.lshift_ext(n), px.lshift_ext(n))
        _same(x.zeroext(n), px.zeroext(n))
        _same(x.signext(n), px.signext(n))
    for m in (0, 1, 0x80000000, 0xFFFFFFFF, 0x9908B0DF, rng.getrandbits(32)):
        _same(x & m, px & m)
        _same(m & x, m & px)
        _same(x | m, px | m)
    _same(x % 256, px % 256)
    with pytest.raises(ValueError):
        px % 3
    _same(x[3], px[3])
    _same(x[-1], px[-1])
    _same(x[4:20], px[4:20])
    _same(x[::-1], px[::-1])
    _same(x.sum(), px.sum())
    _same(x.broadcast(0, 32), px.broadcast(0, 32))
    _same(x.dup(3), px.dup(3))
    _same(x.concat(b), px.concat(pb))
    _same((x & 0xFFFF) | ((c ^ 5) & 0xFFFF0000), (px & 0xFFFF) | ((pc ^ 5) & 0xFFFF0000))
    with pytest.raises(ValueError):
        px | pc
    with pytest.raises(ValueError):
        px ^ pb                                        # different lengths
    with pytest.raises(TypeError):
        px ^ a                                         # representations do not mix
    s = rng.getrandbits(87)
    assert x.evaluate(s) == px.evaluate(s) and d.evaluate(s) == pd.evaluate(s)
    sol = tuple(rng.getrandbits(w) for w in (32, 16, 32, 7))
    assert lin.evaluate(x, sol) == pk.evaluate(px, sol)


def test_zeros_of_the_recovery_examples_are_identical():
    # MT19937, three of the reference's variants (examples/mt.py:49-54) on a short output stream; xoshiro; an LFSR
    for bs, nout in ((32, 40), (17, 40), (1337, 3)):
        rand = random.Random(3142)
        out = [rand.getrandbits(bs) for _ in range(nout)]
        lin, pk = _pair([32] * 624)
        z = [MT19937(lin.gens()).getrandbits(bs) ^ o for o in out[:1]]
        rng_t, rng_p = MT19937(lin.gens()), MT19937(pk.gens())
        zt = [rng_t.getrandbits(bs) ^ o for o in out] + [lin.gens()[0] ^ 0x80000000]
        zp = [rng_p.getrandbits(bs) ^ o for o in out] + [pk.gens()[0] ^ 0x80000000]
        assert lin.get_eqs(zt) == pk.get_eqs(zp)
        assert z[0]._bits == zt[0]._bits
    r = random.Random(1)
    state = [r.getrandbits(64) for _ in range(4)]
    gen = Xoshiro256starstar(list(state))
    outs = [gen() for _ in range(10)]
    lin, pk = _pair([64] * 4)
    st, sp = Xoshiro256starstar(lin.gens()), Xoshiro256starstar(pk.gens())
    zt = [st.step() ^ Xoshiro256starstar.untemper(o) for o in outs]
    zp = [sp.step() ^ Xoshiro256starstar.untemper(o) for o in outs]
    assert lin.get_eqs(zt) == pk.get_eqs(zp)
    assert H.fingerprint(H.padded_eqs(lin, zt), 256) == H.GOLDEN["xoshiro"]["sha256"]


def test_get_rows_drops_zeros_and_takes_bare_ints():
    lin, pk = _pair([8, 8])
    (a, b), (pa, pb) = lin.gens(), pk.gens()
    zeros_t = [a ^ a, a ^ b ^ 3, 0, 5, b & 0]
    zeros_p = [pa ^ pa, pa ^ pb ^ 3, 0, 5, pb & 0]
    assert lin.get_eqs(zeros_t) == pk.get_eqs(zeros_p)
    rows = pk.get_rows(zeros_p)
    assert rows.shape == (len(lin.get_eqs(zeros_t)), 1) and rows.dtype == np.uint64
    assert pk.solve_one([pa ^ pa ^ 1]) is None              # "1 = 0" is decided on the host, like the reference


def test_iteration_indexing_and_negative_constants_follow_the_tuple_front_end():
    """an out-of-range index ends iteration (BitVec has no __iter__: Python walks __getitem__ until
    IndexError); negative constants contribute their MAGNITUDE as the reference's to_bits does (_internal.c:504-531);
    shifts by more than the length grow the vector like bits[n:] + (0,) * n."""
    lin, pk = _pair([8, 5])
    a, b = lin.gens()
    pa, pb = pk.gens()
    assert len(list(pa)) == len(pa) == 8
    assert [x._bits for x in pa] == [x._bits for x in a]
    for i in (-8, -1, 0, 7):
        _same(a[i], pa[i])
    for bad in (8, -9, 100):
        with pytest.raises(IndexError):
            pa[bad]
        with pytest.raises(IndexError):
            a[bad]
    for v in (-1, -2, -0x55, -(1 << 40)):
        _same(v ^ a, v ^ pa)
        _same(a & v, pa & v)
        _same(a | v, pa | v)
        _same(a & ~v, pa & ~v)
    for n in (9, 13):
        _same(a >> n, pa >> n)
        _same(a << n, pa << n)


def test_get_rows_returns_an_array_of_its_own():
    lin, pk = _pair([16])
    (x,) = pk.gens()
    r1 = pk.get_rows([x ^ 0x1234])
    keep = r1.copy()
    r2 = pk.get_rows([x ^ 0xFFFF])
    assert (r1 == keep).all() and not (r1 == r2).all()
