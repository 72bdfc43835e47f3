"""Column-slab sharding of ONE system on real kernels: world size 1 in-process, and two ranks that share
this box's one GPU (gloo moves the per-block payload and the tiles; on a multi-GPU node the same code runs over RCCL).
The sharded solve must give exactly what gf2bv_solve_device gives on the same matrix.  (The schedule over RCCL itself: a child
process, tests/test_gpu_external.py::test_slab_schedule_over_rccl_world_size_1.)"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gf2bv_amd import hip, slab

pytestmark = pytest.mark.gpu


def _system(n, seed, dev):
    stride = hip.padded_stride(n)
    aug = torch.empty(n * stride, dtype=torch.int64, device=dev)
    hip.synth_device(aug.data_ptr(), n, n, stride, seed, device=0, stream=torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize(dev)
    return aug, stride


def _host_system(rows, cols, cap, consistent, seed, dev):
    """rank-deficient / inconsistent systems from the generator of the parity tests, uploaded row-major"""
    import random
    from oracle import gf2_oracle as O
    from tests.systems import random_system
    eqs = random_system(random.Random(seed), rows, cols, .5, cap, consistent, 0)
    stride = hip.padded_stride(cols)
    aug = torch.from_numpy(O.eqs_to_aug(eqs, cols, stride).view(np.int64).reshape(-1)).to(dev)
    return aug, stride


def _worker(rank, world, port, n, seed, q, shape=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda", 0)                      # both ranks on the box's one GPU
        rows = cols = n
        if shape is None:
            aug, stride = _system(n, seed, dev)
        else:
            rows, cols, cap, cons = shape
            aug, stride = _host_system(rows, cols, cap, cons, seed, dev)
        sol = slab.solve_one_sharded(aug, rows, cols, stride, 0)
        if rank == 0:
            ref = hip.solve_device(aug.data_ptr(), rows, cols, stride, 0)
            bad = hip.residual_device(aug.data_ptr(), rows, cols, stride, sol.origin) if sol.solved else 0
            # the CPU oracle on the same words (replaces the single _mzd_pluq call, gf2bv/_internal.c:431-433)
            from oracle import gf2_oracle as O
            host = aug.cpu().numpy().view(np.uint64).reshape(rows, stride)
            want = O.solve_words(np.ascontiguousarray(host), rows, cols, 0, algo=1)
            q.put((sol.status, sol.rank, sol.origin.copy(), sol.pivots.copy(), ref.status, ref.rank, ref.origin.copy(),
                   ref.pivots.copy(), bad, int(want["status"]), int(want["rank"]), np.array(want["origin"]).copy(),
                   np.array(want["pivcols"]).copy()))
        else:
            assert sol is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _run(world, n, seed, shape=None):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, seed, q, shape)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        got = q.get(timeout=200)
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    finally:
        for p in procs:                                    # a worker that is still there (failed partner, hang) must not outlive the test
            if p.is_alive():
                p.kill()
                p.join(timeout=10)
    st, rk, org, piv, rst, rrk, rorg, rpiv, bad, ost, ork, oorg, opiv = got
    assert (st, rk) == (rst, rrk) and bad == 0
    assert np.array_equal(org, rorg) and np.array_equal(piv, rpiv)
    # ... and word for word what the oracle says: status, rank, column rank profile, origin
    assert (st, rk) == (ost, ork) and np.array_equal(piv, opiv)
    if st == 0:
        assert np.array_equal(org, oorg)
    return st, rk


@pytest.mark.parametrize("world,n,seed", [(1, 3000, 11), (2, 4096, 12), (2, 5000, 13), (3, 6200, 14), (4, 8300, 15)])
def test_sharded_solve_equals_single_gpu_solve(world, n, seed):
    _run(world, n, seed)


@pytest.mark.parametrize("K", [2, 3])
def test_slab_handle_ignores_a_forced_two_level_plan(K, monkeypatch):
    """A slab handle is driven block by block (slab_factor_on / slab_apply_on) and never runs outer passes: with
    GF2BV_TWO_LEVEL forced -- or by default from ~68000^2 up -- plan_two_level used to give it outer panels all the same
    (world size 1: no look-ahead at a panel's last block, stale window, silently wrong elimination).  ADVICE round 3."""
    monkeypatch.setenv("GF2BV_TWO_LEVEL", str(K))
    _run(1, 3000, 11)
    _run(2, 4096, 12)


@pytest.mark.parametrize("shape,want_status", [((1500, 1300, 1000, True), 0), ((1500, 1300, 1000, False), 1),
                                               ((2600, 2050, None, True), 0)])
def test_sharded_solve_rank_deficient_and_inconsistent(shape, want_status):
    st, rk = _run(2, 0, 77, shape)
    assert st == want_status and rk <= shape[1]
