"""Small systems in ONE launch (k_small_solve, round 4): every system whose augmented matrix fits the LDS of one workgroup
-- the reference's own examples: README 4 x 4, examples/simple.py 128 x 128, examples/xoshiro.py 640 x 256 -- is solved by
one kernel (elimination to the reduced row echelon form in the LDS, origin / pivots / kernel vectors read off it).  Every
shape here goes down BOTH paths (GF2BV_SMALL=0: the blocked multi-launch path) through all three single-system entry points
(packed host words, CPython digits at 30 and 32 bits, device-resident matrix) and must equal the CPU oracle bit for bit:
status, rank, column rank profile, origin, kernel basis in M4RI's order (gf2bv/_internal.c:309-357, 431-455)."""
import random

import numpy as np
import pytest

from gf2bv_amd import LinearSystem, _internal, hip, m4ri_solve
from oracle import gf2_oracle as O
from tests.systems import random_system, structured_system
from tests.test_gpu_parity import assert_same

pytestmark = pytest.mark.gpu

SMALL_SHAPES = [
    # rows, cols, density, rank_cap, consistent, zero_rows
    (1, 1, .5, None, True, 0), (4, 4, .5, None, True, 1), (8, 5, .5, None, True, 0), (64, 63, .5, None, True, 0),
    (64, 64, .5, None, True, 0), (66, 65, .5, None, True, 0), (128, 127, .5, None, True, 0), (130, 128, .5, None, True, 0),
    (200, 129, .5, 77, True, 0), (300, 200, .5, 40, True, 0), (300, 200, .5, 40, False, 0), (300, 200, .1, None, True, 20),
    (300, 200, .5, 0, True, 300), (640, 256, .05, None, True, 0), (640, 256, .5, None, True, 0), (640, 256, .5, 255, False, 0),
    (800, 767, .5, None, True, 0), (900, 700, .5, 600, True, 30), (1200, 511, .5, 500, True, 0), (2000, 300, .01, None, True, 0),
    (960, 900, .5, None, True, 0), (1410, 704, .01, None, True, 0),          # (too large for the LDS with the tables: blocked path)
    (4096, 100, .5, 70, True, 100), (4096, 191, .5, None, False, 0), (2000, 500, .002, None, True, 0), (700, 640, .5, 639, True, 0),
    (1000, 1000, .5, None, True, 0),          # one word too wide for the LDS: the blocked path whatever GF2BV_SMALL says
]


def _eligible(rows, cols):
    """gf2_solver.hip: small_eligible -- matrix + 2 x 256 table entries at an odd row pitch in 18432 words of LDS"""
    wt = (cols + 1 + 63) // 64
    return cols <= 1023 and rows <= 4096 and (rows + 512) * (wt | 1) <= 18432


def _digits(eqs, bpd):
    nd = [max(1, (abs(e).bit_length() + bpd - 1) // bpd) for e in eqs]
    off = np.zeros(len(eqs) + 1, dtype=np.int64)
    np.cumsum(nd, out=off[1:])
    dig = np.zeros(int(off[-1]), dtype=np.uint32)
    for i, e in enumerate(eqs):
        e, k = abs(e), int(off[i])
        while e:
            dig[k] = e & ((1 << bpd) - 1)
            e >>= bpd
            k += 1
    return dig, off


@pytest.mark.parametrize("shape", SMALL_SHAPES, ids=lambda s: f"{s[0]}x{s[1]}-{s[2]}-{s[3]}-{int(s[4])}")
def test_one_launch_path_equals_oracle_and_blocked_path(shape, monkeypatch):
    rows, cols, dens, cap, cons, zr = shape
    rng = random.Random(hash(shape) & 0xFFFF)
    eqs = random_system(rng, rows, cols, dens, cap, cons, zr)
    rng.shuffle(eqs)
    aug = O.eqs_to_aug(eqs, cols)
    stride = hip.padded_stride(cols)
    wide = np.zeros((rows, stride), dtype=np.uint64)
    wide[:, :aug.shape[1]] = aug
    buf = hip.DeviceBuffer(wide.nbytes)
    buf.upload(wide)
    for mode in (0, 1):
        want = O.solve_words(aug, rows, cols, mode)
        for small in ("1", "0"):
            monkeypatch.setenv("GF2BV_SMALL", small)
            got = [hip.solve_words(aug, rows, cols, mode), hip.solve_device(buf.ptr, rows, cols, stride, mode)]
            for bpd in (30, 32):
                dig, off = _digits(eqs, bpd)
                got.append(hip.solve_digits(dig, off, bpd, rows, cols, mode))
            for g in got:
                assert_same(g, want, mode)
                assert g.stats["small_path"] == (int(small) if _eligible(rows, cols) else 0), (small, g.stats)
    assert np.array_equal(buf.download().reshape(wide.shape), wide)          # the input is never modified
    buf.free()


@pytest.mark.parametrize("kind", ["zero_cols", "dup_cols", "dup_head", "dead_head"])
def test_structured_systems_on_the_one_launch_path(kind):
    """Panels that stay short of pivots, dependent columns, duplicate and dead rows at the head of the scan (the candidate
    passes of a panel then repeat: several rounds of 64 candidates per panel)."""
    rng = random.Random(sum(map(ord, kind)) + 1)
    for rows, cols in ((70, 64), (200, 130), (700, 640), (800, 767), (3000, 129)):
        eqs = structured_system(rng, rows, cols, kind)
        aug = O.eqs_to_aug(eqs, cols)
        for mode in (0, 1):
            got = hip.solve_words(aug, rows, cols, mode)
            assert got.stats["small_path"] == 1
            assert_same(got, O.solve_words(aug, rows, cols, mode), mode)


def test_bits_above_cols_and_the_sign_are_ignored_and_trivial_systems():
    # (gf2bv/_internal.c:41-59, 414: |v|'s digits, bits above cols + 1 ignored)
    cols = 70
    rng = random.Random(5)
    eqs = [rng.getrandbits(cols + 1) for _ in range(90)]
    noisy = [(-1 if i % 3 == 0 else 1) * (e | (rng.getrandbits(40) << (cols + 1))) for i, e in enumerate(eqs)]
    for mode in (0, 1):
        want = O.m4ri_solve(list(eqs), cols, mode)
        got = m4ri_solve(noisy, cols, mode)
        if mode == 0 or want is None:
            assert got == want
        else:
            assert (got.dimension, got.origin, got.basis) == (want.dimension, want.origin, want.basis)
    # all zero: rank 0, every variable free (S4 with r = 0: free = 0 .. cols - 1); 0 = 1: no solution
    for rows, cols in ((5, 5), (130, 70), (300, 256)):
        aug = np.zeros((rows, O.words_for(cols)), dtype=np.uint64)
        got = hip.solve_words(aug, rows, cols, 1)
        assert got.stats["small_path"] == 1 and got.status == 0 and got.rank == 0 and got.dimension == cols
        assert got.origin_int() == 0 and got.basis_ints() == tuple(1 << i for i in range(cols))
        aug[rows - 1, cols // 64] = np.uint64(1 << (cols % 64))
        assert hip.solve_words(aug, rows, cols, 1).status == 1
    # the README system and its hand-derived answer
    sp = m4ri_solve([15, 20, 11, 0], 4, 1)
    assert (sp.origin, sp.basis) == (0b0001, (0b0101,))
    lin = LinearSystem([2, 2])
    x, y = lin.gens()
    assert list(lin.solve_all([x ^ y ^ 1, (x & 1) ^ (y >> 1)])) == [lin.convert_sol(s) for s in
                                                                   O.m4ri_solve(lin.get_eqs([x ^ y ^ 1, (x & 1) ^ (y >> 1)]) + [0, 0], 4, 1)]


def test_device_resident_small_system_is_ordered_after_its_producer():
    """gf2bv_solve_device on a small system: the matrix is whatever the caller's stream (here the null stream) has produced -- the
    one launch goes onto that stream like the blocked path's launches.  A generator kernel and the solve back to back, 60 times,
    new contents each time, against the oracle on the generator's definition."""
    rows, cols = 640, 256
    stride = hip.padded_stride(cols)
    buf = hip.DeviceBuffer(rows * stride * 8)
    for seed in range(60):
        hip.synth_device(buf.ptr, rows, cols, stride, 900 + seed)            # asynchronous, null stream
        got = hip.solve_device(buf.ptr, rows, cols, stride, 1)
        assert got.stats["small_path"] == 1
        assert_same(got, O.solve_words(O.gen_synthetic(rows, cols, 900 + seed), rows, cols, 1), 1)
    buf.free()


def test_many_small_solves_from_many_threads():
    """the staging buffers are per host thread: 12 threads x 40 solves of three shapes, every answer against the oracle"""
    from concurrent.futures import ThreadPoolExecutor
    rng = random.Random(91)
    jobs = []
    for rows, cols, cap in ((640, 256, None), (130, 128, 100), (300, 200, 40), (800, 767, None)):
        eqs = random_system(rng, rows, cols, .5, cap, True, 0)
        jobs.append((eqs, cols, O.m4ri_solve(list(eqs), cols, 1)))

    def run(k):
        eqs, cols, want = jobs[k % len(jobs)]
        got = _internal.m4ri_solve(eqs, cols, 1)
        return (got.dimension, got.origin, got.basis) == (want.dimension, want.origin, want.basis)

    with ThreadPoolExecutor(12) as ex:
        assert all(ex.map(run, range(480)))


def test_warm_latency_of_the_xoshiro_shape():
    """640 x 256 (examples/xoshiro.py): the blocked path needs ~0.26 ms warm; the one-launch path must beat it clearly."""
    import time
    rng = random.Random(3)
    eqs = random_system(rng, 640, 256, .05, None, True, 0)
    want = O.m4ri_solve(list(eqs), 256, 0)
    ts = []
    for _ in range(300):
        t0 = time.perf_counter()
        got = m4ri_solve(eqs, 256, 0)
        ts.append(time.perf_counter() - t0)
    assert got == want
    med = sorted(ts)[len(ts) // 2]
    print(f"warm m4ri_solve 640 x 256: median {med * 1e6:.0f} us, p10 {sorted(ts)[30] * 1e6:.0f} us")
    assert med < 200e-6, med
