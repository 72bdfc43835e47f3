"""Host-side logic: BitVec algebra, LinearSystem plumbing, the _internal helper functions and
AffineSpace get / iteration order.  No GPU (m4ri_solve itself is only checked for its argument
errors and for failing loudly without a device)."""
import pickle
import random

import pytest

from gf2bv_amd import BitVec, DimensionTooLargeError, LinearSystem, _internal
from gf2bv_amd._internal import m4ri_solve, mul_bit_quad, to_bits, tuple_where, xor_tuple
from oracle import gf2_oracle as O


def test_to_bits_xor_tuple_tuple_where():
    assert to_bits(5, 0b10110) == (False, True, True, False, True)
    assert to_bits(3, -5) == (True, False, True)                     # magnitude only (_internal.c:5-16)
    assert to_bits(70, (1 << 65) | 1)[65] is True and len(to_bits(70, 1)) == 70
    assert to_bits(0, 7) == ()
    with pytest.raises(ValueError):
        to_bits(-1, 3)
    with pytest.raises(TypeError):
        to_bits(3, "x")
    assert xor_tuple((1, 2, 3), (3, 2, 1)) == (2, 0, 2)
    with pytest.raises(ValueError):
        xor_tuple((1,), (1, 2))
    with pytest.raises(TypeError):
        xor_tuple([1], (1,))
    cond = tuple([True, False, True])
    ret = tuple_where(cond, (10, 11, 12), 0)
    assert ret is cond and ret == (10, 0, 12)                        # mutates its first argument (_internal.c:672)
    assert tuple_where((0, 1), 7, (8, 9)) == (8, 7)
    with pytest.raises(ValueError):
        tuple_where((1, 0), (1,), 0)


def test_mul_bit_quad_matches_definition():
    rng = random.Random(3)
    n = 6
    basis = [1 << i for i in range(1 + n + n * (n - 1) // 2)]
    for _ in range(50):
        a, b, v = rng.getrandbits(n), rng.getrandbits(n), rng.getrandbits(4)
        want, mi = v, 1 + n
        for i in range(n):
            for j in range(i):
                if ((a >> i) & (b >> j) & 1) ^ ((a >> j) & (b >> i) & 1):
                    want |= basis[mi]
                mi += 1
        assert mul_bit_quad(n, a, b, v, basis) == want
    with pytest.raises(ValueError):
        mul_bit_quad(n, 1, 1, 0, basis[:-1])


def _rand_env(rng, lin):
    sol = tuple(rng.getrandbits(s) for s in lin._sizes)
    return sol


def test_bitvec_ops_commute_with_evaluation():
    """Every BitVec operator, evaluated at a random point, equals the same operator on ints."""
    rng = random.Random(42)
    lin = LinearSystem([32, 32])
    x, y = lin.gens()
    M = (1 << 32) - 1

    def rotl(v, n): return ((v << n) | (v >> (32 - n))) & M
    def rotr(v, n): return ((v >> n) | (v << (32 - n))) & M
    for _ in range(30):
        sol = _rand_env(rng, lin)
        a, b = sol
        k, c = rng.randrange(1, 31), rng.getrandbits(32)
        cases = [
            (x ^ y, a ^ b), (x ^ c, a ^ c), (c ^ x, a ^ c), (x >> k, a >> k), (x << k, (a << k) & M),
            (x & c, a & c), (c & x, a & c), (x | c, a | c), (x % 16, a % 16), (x.rotl(k), rotl(a, k)), (x.rotr(k), rotr(a, k)),
            (x.sum(), bin(a).count("1") & 1), (x[3:9], (a >> 3) & 63), (x[5], (a >> 5) & 1),
            (x.lshift_ext(4), a << 4), (x.zeroext(8), a), (x.concat(y), a | (b << 32)), (x.dup(2), a | (a << 32)),
            (x.broadcast(7, 5), 31 * ((a >> 7) & 1)), (x.signext(3), a | (7 * (a >> 31) << 32)),
            ((x & 0xFF) | ((y & 0xFF) << 0 & 0) , a & 0xFF), ((x ^ (x >> 11) & c) ^ ((y << 7) & 0x9D2C5680), (a ^ (a >> 11) & c) ^ ((b << 7) & 0x9D2C5680 & M)),
        ]
        for bv, want in cases:
            assert lin.evaluate(bv, sol) == want
    assert (x & M) is x and (x >> 0) is x and (x << 0) is x
    with pytest.raises(ValueError):
        x ^ x[:5]
    with pytest.raises(ValueError):
        x % 12
    with pytest.raises(ValueError):
        x | y                                        # both sides symbolic at the same position
    assert len(x | (y & 0).zeroext(4)) == 36


def test_linear_system_plumbing():
    lin = LinearSystem([3, 2])
    a, b = lin.gens()
    assert lin._cols == 5 and a._bits == (2, 4, 8) and b._bits == (16, 32)
    zeros = [a ^ 0b101, 0, b, 7]
    assert lin.get_eqs(zeros) == [3, 4, 9, 16, 32, 7]            # literal zeros dropped, ints kept
    assert lin.solve_one([a ^ a ^ 1]) is None                     # "1 = 0" short-circuits before the GPU
    assert list(lin.solve_all([a ^ a ^ 1])) == []
    assert lin._convert_sol(0b10101) == (0b101, 0b10)
    with pytest.raises(AssertionError):
        lin._convert_sol(1 << 5)
    lin2 = pickle.loads(pickle.dumps(lin))
    assert lin2._sizes == [3, 2] and lin2.gens()[1]._bits == (16, 32)
    assert lin.evaluate(a ^ (b.zeroext(1)), (0b110, 0b11)) == 0b110 ^ 0b011


def test_m4ri_solve_argument_errors():
    with pytest.raises(TypeError, match="requires 3 arguments"):
        m4ri_solve([1, 2], 2)
    with pytest.raises(TypeError, match="must be a list"):
        m4ri_solve((1, 2), 2, 0)
    with pytest.raises(ValueError, match="columns must be positive"):
        m4ri_solve([1, 2], 0, 0)
    with pytest.raises(ValueError, match="Invalid mode"):
        m4ri_solve([1, 2], 2, 5)
    with pytest.raises(ValueError, match="greater than or equal"):
        m4ri_solve([1], 2, 0)
    with pytest.raises(TypeError, match="must be integers"):
        m4ri_solve([1, "x"], 2, 0)


def test_m4ri_solve_many_argument_errors():
    many = _internal.m4ri_solve_many
    with pytest.raises(TypeError, match="requires 3 arguments"):
        many([[1, 2]], 2)
    with pytest.raises(TypeError, match="list of equation lists"):
        many([(1, 2)], 2, 0)
    with pytest.raises(ValueError, match="columns must be positive"):
        many([[1, 2]], 0, 0)
    with pytest.raises(ValueError, match="Invalid mode"):
        many([[1, 2]], 2, 2)
    with pytest.raises(ValueError, match="same number of rows"):
        many([[1, 2], [1, 2, 3]], 2, 0)
    with pytest.raises(ValueError, match="greater than or equal"):
        many([[1]], 2, 0)
    with pytest.raises(TypeError, match="must be integers"):
        many([[1, 2], [1, None]], 2, 0)
    assert many([], 3, 1) == []
    with pytest.raises(TypeError, match="requires 3 arguments"):
        many([[1, 2]], 2, 0, None, 1)
    with pytest.raises(ValueError, match="must not be empty"):
        many([[1, 2]], 2, 0, [])
    with pytest.raises(ValueError, match="out of range"):
        many([[1, 2]], 2, 0, [-1])
    with pytest.raises(TypeError):
        many([[1, 2]], 2, 0, ["gpu0"])
    with pytest.raises(ValueError, match="'all'"):           # devices: None (default device) | "all" | int | sequence of ints
        many([[1, 2]], 2, 0, "gpu0")
    # "1 = 0" systems are decided on the host, the others would go to the GPU together
    lin = LinearSystem([2])
    (v,) = lin.gens()
    assert lin.solve_one_many([[1], [v ^ v, 1]]) == [None, None]


def test_device_selection_surface():
    """round 3: every solve entry takes an optional trailing device; the module keeps a default (GF2BV_DEVICE or 0)"""
    assert _internal.get_default_device() == 0
    with pytest.raises(ValueError, match="out of range"):
        _internal.set_default_device(-3)
    with pytest.raises(ValueError, match="out of range"):
        _internal.m4ri_solve([1, 2], 2, 0, -1)
    with pytest.raises(TypeError, match="requires 3 arguments"):
        _internal.m4ri_solve([1, 2], 2, 0, 0, 0)
    sp = _internal._space_from_ints(4, 1, (5,))
    assert sp.device == -1                       # host-built: walked on the host


def test_planted_solution_of_the_synthetic_generator_matches_the_oracle_definition():
    """hip.planted_solution (the product's own statement of the generator's planted vector: what bench.py and the large-size
    tests compare a full-rank solve_one with) against the oracle's C definition."""
    import numpy as np

    from gf2bv_amd import hip
    for cols, seed in ((1, 0), (63, 1), (64, 5), (65, 5), (777, 99), (65536, 1234), (262144, 1242)):
        assert np.array_equal(hip.planted_solution(cols, seed), O.planted_solution(cols, seed)), (cols, seed)


def test_solve_fails_loudly_without_gpu():
    if _internal.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError, match="no HIP device"):
        m4ri_solve([3, 5], 2, 0)
    lin = LinearSystem([2])
    (v,) = lin.gens()
    with pytest.raises(RuntimeError):
        lin.solve_one([v ^ 1])


def test_affine_space_matches_reference_orders():
    """AffineSpace.get / __iter__ (gf2bv/_internal.c:101-122, 63-91, 242-273) vs the oracle's restatement."""
    rng = random.Random(9)
    for cols, dim in ((4, 1), (70, 5), (200, 0), (130, 12), (300, 66)):
        origin = rng.getrandbits(cols)
        basis = tuple(rng.getrandbits(cols) | 1 for _ in range(dim))
        sp = _internal._space_from_ints(cols, origin, basis)
        ref = O.OracleSpace(origin, basis)
        assert sp.dimension == dim and sp.origin == origin and sp.basis == basis
        assert isinstance(sp.basis, tuple)
        n = min(2 ** dim, 300)
        it, rit = iter(sp), iter(ref)
        assert [next(it) for _ in range(n)] == [next(rit) for _ in range(n)]
        assert type(it).__name__ == ("AffineSpaceIterator" if dim <= 64 else "AffineSpaceIteratorSlow")
        for k in (0, 1, 2, 5, (1 << dim) - 1 if dim else 0, 1 << (dim + 3)):
            assert sp.get(k) == ref.get(k)
    sp = _internal._space_from_ints(4, 1, (5,))
    assert list(sp) == [1, 4] and list(sp) == [1, 4]              # a fresh iterator every time
    with pytest.raises(TypeError):
        sp.get()
    with pytest.raises(TypeError):
        sp.get("1")
    with pytest.raises(TypeError):
        _internal.AffineSpace()


def test_dimension_guard(monkeypatch):
    import gf2bv_amd.linsys as L
    fake = _internal._space_from_ints(40, 0, tuple(1 << i for i in range(20)))
    monkeypatch.setattr(L, "m4ri_solve", lambda eqs, cols, mode: fake)
    lin = LinearSystem([40])
    with pytest.raises(DimensionTooLargeError) as ei:
        next(lin.solve_all([lin.gens()[0] & 0]))
    assert ei.value.space is fake
    small = _internal._space_from_ints(40, 3, tuple(1 << i for i in range(2, 12)))
    monkeypatch.setattr(L, "m4ri_solve", lambda eqs, cols, mode: small)
    sols = list(lin.solve_all([], max_dimension=10))
    assert len(sols) == len(set(sols)) == 1024 and sols[0] == (3,) and sols[1] == (3 ^ 4,)
