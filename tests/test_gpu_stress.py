"""Seeded, bounded slices of the open-ended differential runs (tests/manual/stress_parity.py) where the machinery is thinnest
(VERDICT round 2, item 6): kernel bases of 8-63 vectors with rows >> cols, mode-1 gangs of mixed rank at 8192^2, the
two-level elimination forced onto small systems, and the "gate timed out -> void solve -> repeat with events" path."""
import os
import random
import sys

import numpy as np
import pytest

from gf2bv_amd import hip
from oracle import gf2_oracle as O
from tests.child import run_child
from tests.systems import random_system

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _same(got, want, mode):
    assert got.status == want["status"] and got.rank == want["rank"]
    assert np.array_equal(got.pivots, want["pivcols"][: want["rank"]])
    if want["status"] == 0:
        assert np.array_equal(got.origin, want["origin"])
        if mode == 1:
            assert got.dimension == want["dim"]
            assert np.array_equal(got.basis.reshape(-1), np.asarray(want["basis"]).reshape(-1))


@pytest.mark.parametrize("dim", [8, 9, 17, 31, 40, 63])
def test_kernel_bases_of_8_to_63_vectors_rows_much_larger_than_cols(dim):
    """solve_all with a kernel of `dim` vectors: the parity back-substitution in groups of 8 right-hand sides (dim + 1 of
    them), 3-5 x as many equations as unknowns, dense and sparse."""
    rng = random.Random(1000 + dim)
    for cols, density in ((700 + dim, .5), (1500 + 3 * dim, .05)):
        rows = cols * rng.choice([3, 4, 5]) + rng.randint(0, 63)
        eqs = random_system(rng, rows, cols, density, cols - dim, True, rng.randint(0, 40))
        rng.shuffle(eqs)
        aug = O.eqs_to_aug(eqs, cols)
        want = O.solve_words(aug, rows, cols, 1)
        assert want["status"] == 0 and want["dim"] == dim
        _same(hip.solve_words(aug, rows, cols, 1), want, 1)


@pytest.mark.timeout(600)
def test_mode1_gang_of_mixed_ranks_at_8192(monkeypatch):
    """One gang, mode 1 (kernel bases), 8192 x 8192: full rank, 1 / 5 / 70 short of it, inconsistent, sparse -- lock-step launches
    with per-system state; every member against the oracle."""
    monkeypatch.setenv("GF2BV_GANG", "6")
    n = 8192
    rng = random.Random(8192)
    specs = [(.5, None, True), (.5, n - 1, True), (.5, n - 5, True), (.5, n - 70, True), (.5, n - 3, False), (.01, None, True)]
    augs = np.stack([O.eqs_to_aug(random_system(rng, n, n, d, cap, cons, 0), n) for d, cap, cons in specs])
    got = hip.solve_batch_words(augs, n, n, 1)
    assert all(g.stats["gang_systems"] == 6 and g.stats["handover_retries"] == 0 for g in got)
    for a, g in zip(augs, got):
        _same(g, O.solve_words(a, n, n, 1), 1)


@pytest.mark.parametrize("fused", [1, 0])
def test_fused_search_and_narrow_launch(monkeypatch, fused):
    """Round 3's panel path: k_block_fast_narrow (workgroup 0 searches, the others narrow every panel the moment its pivot
    rows are formed -- progress counter in SolveState, write-through stores for everything the bulk stream reads) against the
    two launches it replaces (GF2BV_FUSED_NARROW=0) and the oracle: full rank, rank caps in the middle of a block (the search gives up after it has
    published some panels: poison, resume with the general steps), rows >> cols, a gang, and 16 solves in flight at once."""
    from concurrent.futures import ThreadPoolExecutor
    monkeypatch.setenv("GF2BV_FUSED_NARROW", str(fused))
    rng = random.Random(77)
    jobs = []
    for rows, cols, density, cap, cons in ((2700, 2600, .5, None, True), (5000, 4097, .5, 2600, True), (2300, 2200, .5, 2193, True),
                                           (9000, 1300, .5, 700, False), (3100, 3000, .5, 300, True)):
        eqs = random_system(rng, rows, cols, density, cap, cons, 0)
        aug = O.eqs_to_aug(eqs, cols)
        want = O.solve_words(aug, rows, cols, 1)
        got = hip.solve_words(aug, rows, cols, 1)
        _same(got, want, 1)
        jobs.append((aug, rows, cols, want))
    assert got.stats["fast_blocks"] > 0 and got.stats["handover_retries"] == 0
    with ThreadPoolExecutor(16) as ex:
        res = list(ex.map(lambda j: hip.solve_words(j[0], j[1], j[2], 1), jobs * 6))
    for j, g in zip(jobs * 6, res):
        _same(g, j[3], 1)
    n = 2048
    augs = np.stack([O.eqs_to_aug(random_system(rng, n, n, .5, cap, True, 0), n) for cap in (None, n - 1, 1500, None, 700)])
    monkeypatch.setenv("GF2BV_GANG", "5")
    for a, g in zip(augs, hip.solve_batch_words(augs, n, n, 1)):
        _same(g, O.solve_words(a, n, n, 1), 1)


@pytest.mark.parametrize("K,chain", [(2, 0), (3, 0), (4, 0), (8, 0), (12, 0), (4, 1), (8, 1), (12, 1)])
def test_two_level_elimination_forced_on_small_systems(monkeypatch, K, chain):
    """GF2BV_TWO_LEVEL=K: outer panels of K = 2 ... 12 blocks from the first block on (k_outer_trsm<IDENT> + k_outer_apply + k_update16k; chain:
    GF2BV_OUTER_CHAIN=1, the pivot rows by the chain of panel steps on every word group instead of P = T x S) at sizes the oracle
    solves in seconds -- full rank, rank caps inside / at the edge of an outer panel, zero and duplicate-heavy rows, rows >> cols,
    inconsistent systems, both modes, events and flags."""
    monkeypatch.setenv("GF2BV_TWO_LEVEL", str(K))
    monkeypatch.setenv("GF2BV_OUTER_CHAIN", str(chain))
    rng = random.Random(40 + K)
    shapes = [(3000, 2500, .5, None, True, 0), (3000, 2500, .5, 1000, True, 0), (3000, 2500, .5, 256 * K, True, 0),
              (3000, 2500, .5, 256 * K + 1, False, 0), (5000, 4097, .5, 4000, True, 300), (9000, 2049, .003, None, True, 0),
              (2600, 2600, .5, 2599, True, 0), (6000, 1537, .1, 700, True, 50), (4200, 4100, .02, None, True, 0)]
    for i, (rows, cols, density, cap, cons, zr) in enumerate(shapes):
        eqs = random_system(rng, rows, cols, density, cap, cons, zr)
        if i % 2:
            rng.shuffle(eqs)
        aug = O.eqs_to_aug(eqs, cols)
        monkeypatch.setenv("GF2BV_FLAG_SYNC", "0" if i % 3 == 2 else "1")
        for mode in ((0, 1) if i < 4 else (i % 2,)):
            _same(hip.solve_words(aug, rows, cols, mode), O.solve_words(aug, rows, cols, mode), mode)
    # the default plan (no forcing) on a system large enough to take outer panels is covered by test_target_262144_properties


@pytest.mark.parametrize("K,side", [(3, 1), (4, 0), (12, 0), (2, 1), (8, 1), (12, 1)])
def test_outer_pass_item_shapes(monkeypatch, K, side):
    """The outer pass (k_update16k_wide: sixteen wavefronts x 12 segments under a budget of 120 registers, items of 12288 rows walked
    chunk-major) against systems of 2600-26000 rows -- one ragged item, several items, rows >> cols --, rank caps, inconsistent systems,
    both modes, with the outer step on the next panel's tiles beside the pass on the inner bulk stream (GF2BV_OUTER_SIDE=1, default) and
    in front of it (0): the oracle's answers.  (Round 5 shipped four workgroup shapes, two item orders and a two-stream split behind
    knobs; their A/B is recorded in profiles/r05_outer_shapes.txt and the knobs went in round 6.)"""
    monkeypatch.setenv("GF2BV_OUTER_SIDE", str(side))
    monkeypatch.setenv("GF2BV_TWO_LEVEL", str(K))
    rng = random.Random(500 + K)
    shapes = [(3000, 2500, .5, None, True, 0), (13000, 2100, .5, None, True, 0), (26000, 1500, .3, 1100, True, 40),
              (5000, 4097, .5, 4000, False, 300), (2600, 2600, .5, 2599, True, 0), (12288, 1300, .5, None, True, 0)]
    for i, (rows, cols, density, cap, cons, zr) in enumerate(shapes):
        eqs = random_system(rng, rows, cols, density, cap, cons, zr)
        if i % 2:
            rng.shuffle(eqs)
        aug = O.eqs_to_aug(eqs, cols)
        mode = i % 2
        _same(hip.solve_words(aug, rows, cols, mode), O.solve_words(aug, rows, cols, mode), mode)


@pytest.mark.parametrize("K", [4, 12])
def test_outer_panel_with_pivotless_blocks_in_the_middle(monkeypatch, K):
    """An outer panel whose MIDDLE blocks have no pivot at all (512 all-zero columns, then columns that do have pivots): the outer
    pass skips those blocks and must build the next block's tables from THAT block's pivot rows.  Forced two-level plans, against the
    oracle."""
    monkeypatch.setenv("GF2BV_TWO_LEVEL", str(K))
    rng = random.Random(900 + K)
    rows, cols = 3400, 3000
    eqs = random_system(rng, rows, cols, .5, None, True, 0)
    keep = ~(((1 << 512) - 1) << 256)                  # columns 256 .. 767 go; constants from a planted solution again
    plant = rng.getrandbits(cols)
    coeffs = [(e >> 1) & keep for e in eqs]
    eqs = [(a << 1) | (bin(a & plant).count("1") & 1) for a in coeffs]
    aug = O.eqs_to_aug(eqs, cols)
    for mode in (0, 1):
        want = O.solve_words(aug, rows, cols, mode)
        assert want["rank"] <= cols - 512
        _same(hip.solve_words(aug, rows, cols, mode), want, mode)


@pytest.mark.parametrize("sparse", ["1", "0"])
def test_sparse_block_search(monkeypatch, sparse):
    """Round 5: systems whose first block the dense one-launch search cannot take go through k_block_sparse -- the pool = alive rows
    with a non-zero window (bit masks left by the look-ahead), a parallel selection of 64 independent rows per panel, then the dense
    body.  Sparse systems of 0.1-3 % density with rows >= cols and rows >> cols, shuffled, with rank caps (a panel the pool cannot
    complete: the search gives up, the host resumes that block with the general steps and goes on sparse), inconsistent ones, both
    modes, flags and events; GF2BV_SPARSE_FAST=0: the same systems through the general steps only.  Against the oracle."""
    monkeypatch.setenv("GF2BV_SPARSE_FAST", sparse)
    rng = random.Random(77)
    shapes = [(5000, 4600, .002, None, True, 0), (9000, 4100, .004, None, True, 0), (6000, 5200, .01, None, True, 100),
              (5200, 5000, .03, None, True, 0), (7000, 4200, .003, 3000, True, 0), (7000, 4200, .003, 3000, False, 0),
              (12000, 3100, .001, None, True, 0), (4700, 4600, .0015, None, True, 0)]
    took = 0
    for i, (rows, cols, density, cap, cons, zr) in enumerate(shapes):
        eqs = random_system(rng, rows, cols, density, cap, cons, zr)
        if i % 2 == 0:
            rng.shuffle(eqs)
        aug = O.eqs_to_aug(eqs, cols)
        monkeypatch.setenv("GF2BV_FLAG_SYNC", "0" if i % 3 == 2 else "1")
        for mode in ((0, 1) if i < 3 else (i % 2,)):
            got = hip.solve_words(aug, rows, cols, mode)
            _same(got, O.solve_words(aug, rows, cols, mode), mode)
            took += got.stats["fast_blocks"]
    assert took > 0 or sparse == "0"           # (GF2BV_SPARSE_FAST=0: the dense search may still take the densest systems' later blocks)


@pytest.mark.parametrize("knob,value", [("GF2BV_STREAM_PAIRS", "0"), ("GF2BV_PLAIN", "1"), ("GF2BV_STREAM_PAIRS", "1")])
def test_stream_pair_knobs_change_nothing_but_speed(monkeypatch, knob, value):
    """Round 5: a single solve's bulk stream is probed to run beside its panel stream (Pool::low_stream_for; GF2BV_STREAM_PAIRS=0 or
    GF2BV_PLAIN=1: any idle one), a batch call's gangs take streams of classes of their own.  Whatever the streams: the oracle's answers,
    single solves before and after a batch call."""
    monkeypatch.setenv(knob, value)
    monkeypatch.setenv("GF2BV_GANG", "3")
    rng = random.Random(2718)
    rows, cols = 3000, 2900
    systems = [random_system(rng, rows, cols, d, cap, cons, 0) for cap, d, cons in
               [(None, .5, True), (None, .004, True), (1500, .5, True), (None, .5, False), (cols - 1, .01, True)]]
    augs = np.stack([O.eqs_to_aug(e, cols) for e in systems])
    for mode in (0, 1):
        want = [O.solve_words(a, rows, cols, mode) for a in augs]
        for a, w in zip(augs[:2], want[:2]):
            _same(hip.solve_words(a, rows, cols, mode), w, mode)
        for g, w in zip(hip.solve_batch_words(augs, rows, cols, mode), want):
            _same(g, w, mode)
        for a, w in zip(augs[1:3], want[1:3]):
            _same(hip.solve_words(a, rows, cols, mode), w, mode)


@pytest.mark.parametrize("inv", ["1", "0"])
def test_back_substitution_with_inverted_diagonal_blocks(monkeypatch, inv):
    """Round 4: from four groups of 16 panels up the back-substitution inverts the diagonal blocks up front (k_bs_inv) and a link
    of its chain is one launch (k_bs_step: far part, then X = Minv x a by the last workgroup to arrive) -- GF2BV_BS_INV=1 forces
    that path onto every size, =0 the two-launch chain with the serial walk (k_bs_far + k_bs_near).  Shapes with a short first
    group, one group, rank caps (panels with few or no pivots), rows >> cols, inconsistent systems, kernel bases of 0 ... 40
    vectors (several passes of 8 right-hand sides), both modes, against the oracle."""
    monkeypatch.setenv("GF2BV_BS_INV", inv)
    monkeypatch.setenv("GF2BV_SMALL", "0")
    rng = random.Random(99)
    shapes = [(70, 64, .5, None, True, 0), (1100, 1023, .5, 900, True, 0), (1100, 1025, .5, None, True, 0), (2100, 2048, .5, None, True, 0),
              (4200, 4100, .5, 4060, True, 0), (5000, 4097, .5, 2600, True, 100), (9000, 2049, .003, None, True, 0), (6000, 5200, .5, 5199, False, 0),
              (8300, 8200, .5, None, True, 0), (3000, 2900, .5, 64 * 17 + 3, True, 0)]
    for i, (rows, cols, density, cap, cons, zr) in enumerate(shapes):
        eqs = random_system(rng, rows, cols, density, cap, cons, zr)
        aug = O.eqs_to_aug(eqs, cols)
        for mode in (0, 1):
            _same(hip.solve_words(aug, rows, cols, mode), O.solve_words(aug, rows, cols, mode), mode)


@pytest.mark.external
@pytest.mark.timeout(600)
def test_soak_many_solves_in_flight():
    """A reduced run of tests/manual/soak_concurrent.py in the automated suite: 2 rounds of 40 solves from 16
    threads -- more solves in flight than the runtime has hardware queues -- over ten shapes with rank caps, every result against
    the oracle.  The in-kernel hand-overs are timing-dependent: k_block_fast_narrow's progress counter (sc1 write-through stores
    of the data, vmcnt(0), then the counter; sc1 loads behind the counter on the consumer side: the second visibility recipe of
    MI355X_MICROARCH.md, no L2 write-back / invalidate beside a running bulk update), the stream gates."""
    out = run_child([sys.executable, os.path.join(ROOT, "tests", "manual", "soak_concurrent.py"), "2", "16", "4711"], 400)
    assert out.ok and "SOAK ok" in out.out, out.report()


_RETRY_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, %r)
from gf2bv_amd import hip
from oracle import gf2_oracle as O
n = 4096
aug = O.gen_synthetic(n, n, 77)
want = O.solve_words(aug, n, n, 0, algo=1)
import time
for k in range(2):
    t = time.time(); got = hip.solve_words(aug, n, n, 0); dt = time.time() - t
    ok = got.status == want["status"] and got.rank == want["rank"] and np.array_equal(got.origin, want["origin"])
    print("SOLVE", k, "ok" if ok else "WRONG", round(dt, 2), flush=True)
"""


@pytest.mark.external
@pytest.mark.timeout(300)
def test_expired_gate_voids_the_solve_and_it_is_repeated_with_events(tmp_path):
    """GF2BV_FLAG_SYNC=2 skips the concurrency probe; under `rocprofv3 --pmc` kernels of two streams do not execute
    concurrently, so the first hand-over gate expires (5 s), the solve is voided, the device marked and the C-ABI entry
    repeats it with events: the FIRST call is slow and right, the next one fast and right (gf2_solver.hip: guarded /
    GF2BV_RETRY_EVENTS; tools/_probe/retry_check.sh was the only cover in round 2)."""
    script = tmp_path / "retry.py"
    script.write_text(_RETRY_SCRIPT % ROOT)
    env = dict(os.environ, GF2BV_FLAG_SYNC="2", TMPDIR="/tmp")
    cmd = ["rocprofv3", "--pmc", "SQ_WAVES", "--kernel-trace", "--output-format", "csv", "-d", str(tmp_path / "prof"), "--",
           sys.executable, str(script)]
    out = run_child(cmd, 200, env=env, cwd="/tmp")
    lines = [ln.split() for ln in out.out.splitlines() if ln.startswith("SOLVE")]
    assert len(lines) == 2, out.report()
    assert lines[0][2] == "ok" and lines[1][2] == "ok"
    assert float(lines[0][3]) > 4.0 and float(lines[1][3]) < 2.0, lines          # one expired gate, then events
