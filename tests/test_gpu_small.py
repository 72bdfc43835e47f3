"""Systems that fit the LDS take ONE launch (k_small_solve: one wavefront, Gauss-Jordan in LDS) instead of the blocked
elimination -- the reference's own examples are that small (README 4 x 4, examples/simple.py 128 x 128, examples/xoshiro.py
640 x 256).  Every entry point, both modes, against the oracle and against the blocked path (GF2BV_SMALL=0)."""
import random
import time

import numpy as np
import pytest

from gf2bv_amd import LinearSystem, _internal, hip, m4ri_solve
from oracle import gf2_oracle as O
from tests import harness as H
from tests.systems import random_system

pytestmark = pytest.mark.gpu


def _same(got, want, mode):
    assert got.status == want["status"] and got.rank == want["rank"]
    assert np.array_equal(got.pivots, want["pivcols"][: want["rank"]])
    if want["status"] == 0:
        assert np.array_equal(got.origin, want["origin"])
        if mode == 1:
            assert got.dimension == want["dim"]
            assert np.array_equal(got.basis.reshape(-1), np.asarray(want["basis"]).reshape(-1))


SMALL_SHAPES = [
    # rows, cols, density, rank_cap, consistent, zero_rows -- up to the limits: 2048 rows, 511 columns, rows x words <= 16384
    (1, 1, .5, None, True, 0), (4, 4, .5, None, True, 1), (8, 5, .5, None, True, 0), (64, 63, .5, None, True, 0),
    (64, 64, .5, None, True, 0), (66, 65, .5, None, True, 0), (128, 127, .5, None, True, 0), (130, 128, .5, None, True, 0),
    (200, 129, .5, 77, True, 0), (300, 200, .5, 40, True, 0), (300, 200, .5, 40, False, 0), (300, 200, .1, None, True, 20),
    (300, 200, .5, 0, True, 300), (640, 256, .05, None, True, 0), (2048, 447, .5, None, True, 0), (2047, 511, .5, 300, True, 5),
    (2048, 511, .02, None, False, 0), (1500, 320, .5, 319, True, 0), (512, 511, .5, None, True, 0), (700, 383, .5, 100, False, 7),
]


@pytest.mark.parametrize("shape", SMALL_SHAPES, ids=lambda s: f"{s[0]}x{s[1]}")
def test_one_launch_solve_matches_oracle_and_blocked_path(shape, monkeypatch):
    rows, cols, dens, cap, cons, zr = shape
    rng = random.Random(hash(shape) & 0xFFFF)
    eqs = random_system(rng, rows, cols, dens, cap, cons, zr)
    rng.shuffle(eqs)
    aug = O.eqs_to_aug(eqs, cols)
    for mode in (0, 1):
        want = O.solve_words(aug, rows, cols, mode)
        monkeypatch.setenv("GF2BV_SMALL", "1")
        got = hip.solve_words(aug, rows, cols, mode)                      # words entry (row-major source)
        assert got.stats["n_sweeps"] == 0 and got.stats["fast_blocks"] == 0
        _same(got, want, mode)
        monkeypatch.setenv("GF2BV_SMALL", "0")
        _same(hip.solve_words(aug, rows, cols, mode), want, mode)           # the blocked path on the same words
    # digits entry (the list-of-int boundary: tile-major source), junk above the RHS column and negative ints included
    monkeypatch.setenv("GF2BV_SMALL", "1")
    noisy = [(-e if i % 3 == 0 else e) | (rng.getrandbits(20) << (cols + 1)) for i, e in enumerate(eqs)]
    for mode in (0, 1):
        got, want = m4ri_solve(list(noisy), cols, mode), O.m4ri_solve(list(eqs), cols, mode)
        if mode == 0 or want is None:
            assert got == want
        else:
            assert (got.dimension, got.origin, got.basis) == (want.dimension, want.origin, want.basis)


def test_device_resident_small_system_and_the_reference_examples(monkeypatch):
    monkeypatch.setenv("GF2BV_SMALL", "1")
    rows, cols = 640, 256
    eqs = random_system(random.Random(9), rows, cols, .05, None, True, 0)
    stride = hip.padded_stride(cols)
    aug = O.eqs_to_aug(eqs, cols, stride)
    buf = hip.DeviceBuffer(aug.nbytes)
    buf.upload(aug)
    _same(hip.solve_device(buf.ptr, rows, cols, stride, 1), O.solve_words(aug, rows, cols, 1), 1)
    assert np.array_equal(buf.download().reshape(aug.shape), aug)            # input untouched
    buf.free()
    # README hand KAT, golden fixtures of the reference's examples through LinearSystem
    sp = m4ri_solve([15, 20, 11, 0], 4, 1)
    assert (sp.dimension, sp.origin, sp.basis, list(sp)) == (1, 0b0001, (0b0101,), [1, 4])
    # ... and the reference's examples through LinearSystem, as tests/test_gpu_parity.py runs them on the blocked path
    G = H.GOLDEN
    e = G["readme4"]
    lin = LinearSystem(e["sizes"])
    a, b, c, d = lin.gens()
    zeros = [a ^ b ^ c ^ 1, b ^ d, a ^ c ^ 1]
    assert [list(s) for s in lin.solve_all(zeros)] == e["expect"]["solve_all"]
    assert list(lin.solve_one(zeros)) == e["expect"]["solve_one"]
    for inp in (None, tuple(int(v, 16) for v in G["simple_affine"]["input"])):
        lin, zeros, expected = H.simple_system(inp)
        sols = list(lin.solve_all(zeros))
        ref = O.m4ri_solve(H.padded_eqs(lin, zeros), 128, 1)
        assert sols == [lin.convert_sol(s) for s in ref]                      # same set AND same Gray order
        assert all(H.magic(*s) == tuple(expected) for s in sols)
    lin, zeros, state, outs = H.xoshiro_system(1, 10)
    assert list(lin.solve_all(zeros)) == [state] and lin.solve_one(zeros) == state


def test_one_launch_latency(monkeypatch):
    """what the path is for: a warm 640 x 256 solve through the words entry takes a fraction of the blocked path's time"""
    rows, cols = 640, 256
    aug = O.eqs_to_aug(random_system(random.Random(3), rows, cols, .05, None, True, 0), cols)

    def best(n=30):
        hip.solve_words(aug, rows, cols, 0)
        ts = []
        for _ in range(n):
            t = time.perf_counter(); hip.solve_words(aug, rows, cols, 0); ts.append(time.perf_counter() - t)
        return min(ts)

    monkeypatch.setenv("GF2BV_SMALL", "1")
    fast = best()
    monkeypatch.setenv("GF2BV_SMALL", "0")
    slow = best()
    print(f"640 x 256 solve_one, warm, best of 30: one launch {fast * 1e6:.0f} us, blocked {slow * 1e6:.0f} us")
    assert fast < slow
