"""The N>1 path on CPU: two gloo ranks shard a batch of independent systems, each solves its
block (here with the CPU oracle standing in for the GPU -- this is a test of sharding + the
single end-of-job gather, not of the solver) and the gathered records must equal the
single-process answer in system order."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gf2bv_amd import batch


def test_shard_bounds_partition():
    for nsys in (0, 1, 5, 8, 13, 512):
        for world in (1, 2, 3, 8):
            blocks = [batch.shard_bounds(nsys, world, r) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == nsys
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in blocks]
            assert max(sizes) - min(sizes) <= 1


def _solve_block(n, seeds):
    from oracle import gf2_oracle as O
    recs = []
    for s in seeds:
        aug = O.gen_synthetic(n, n, s)
        r = O.solve_words(aug, n, n, 0)
        recs.append(batch.make_record(r["status"], r["rank"], r["origin"]))
    return np.stack(recs) if recs else np.zeros((0, batch.record_words(n)), dtype=np.int64)


def _worker(rank, world, port, n, nsys, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = batch.shard_bounds(nsys, world, rank)
        local = torch.from_numpy(_solve_block(n, [1000 + i for i in range(lo, hi)]))
        allrec = batch.gather_records(local, nsys)
        if rank == 0:
            q.put(allrec.numpy().copy())
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("nsys", [5, 4])
def test_two_rank_gather_matches_single_process(nsys):
    n, world = 192, 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, nsys, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = _solve_block(n, [1000 + i for i in range(nsys)])
    assert got.shape == want.shape and np.array_equal(got, want)
    assert (got[:, 0] == 0).all()


@pytest.mark.parametrize("world,total", [(8, 512), (4, 512), (8, 509)])
def test_bench_dry_run_ranks_world_8_real_numbers(world, total):
    """Round 5: `bench.py --gpus 8 --dry-run-ranks` under torch.distributed.run with gloo -- every rank's plan of
    the configs[3] job (512 x 32768^2) without a GPU: the blocks tile the job in rank order, the gangs tile each block, the record
    rows of the one all_gather do not overlap, and what a rank keeps resident (inputs + three gangs' working copies + the pool's
    cap) fits one MI355X."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", str(world), "--dry-run-ranks", "--batch-total", str(total)]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=root)
    assert res.returncode == 0, res.stderr[-2000:]
    d = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["blocks_tile_the_job"] and d["gangs_tile_each_block"] and d["fits_in_hbm"] and len(d["ranks"]) == world
    rows = [tuple(p["record_rows_in_gathered_table"]) for p in d["ranks"]]
    assert rows == [batch.shard_bounds(total, world, r) for r in range(world)]
    for p in d["ranks"]:
        n = p["systems"][1] - p["systems"][0]
        assert p["device"] == p["rank"] and p["gang_size"] >= 8 and p["gang_size"] % 8 == 0      # a system per XCD needs multiples of 8
        assert sum(b - a for a, b in p["gangs"]) == n and p["record_bytes"] == n * batch.record_words(32768) * 8
