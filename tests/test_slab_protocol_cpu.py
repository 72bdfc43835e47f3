"""The column-slab schedule (gf2bv_amd.slab.run_schedule: per-block owner, ONE broadcast of the block's records, every
rank applies the block to its own columns, tiles collected on rank 0) with two gloo ranks on the CPU.  A small integer
engine stands in for the HIP engine on each rank -- Gauss-Jordan on the block's columns by their owner, the payload =
pivot rows + per-row multipliers -- and the collected result is checked against the CPU oracle.  (The HIP engine under
the same schedule: tests/test_gpu_slab.py.)"""
import os
import random
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gf2bv_amd import slab
from tests.systems import random_system

BLOCK = 8            # columns per block
TILE = 16            # columns per tile: two blocks share an owner, like the 8-word tiles of the HIP engine


class IntSlabEngine:
    """rows as Python ints restricted to the columns this rank owns (tile t of TILE columns belongs to rank t % world);
    column `cols` (the right-hand side) lives in the last tile."""

    def __init__(self, eqs, cols, world, rank):
        self.cols, self.world, self.rank = cols, world, rank
        self.width = cols + 1
        self.ntiles = (self.width + TILE - 1) // TILE
        self.nblocks = (cols + BLOCK - 1) // BLOCK
        self.rows = len(eqs)
        full = [((e >> 1) & ((1 << cols) - 1)) | ((e & 1) << cols) for e in eqs]
        self.own = sum(((1 << TILE) - 1) << (TILE * t) for t in range(self.ntiles) if t % world == rank)
        self.A = [r & self.own for r in full]
        self.alive = [True] * self.rows
        self.pivots = []                                   # (column, physical row)
        # fixed-size payload: per block up to BLOCK pivots (column, row), then one multiplier mask per row
        self.payload = torch.zeros(1 + 2 * BLOCK + self.rows, dtype=torch.int64)

    def owner(self, b):
        return ((b * BLOCK) // TILE) % self.world

    def factor(self, b):
        """owner: eliminate the block's columns among the alive rows (on a scratch copy of those columns) and record,
        for every row, which of the block's pivot rows it has to absorb"""
        lo, hi = b * BLOCK, min((b + 1) * BLOCK, self.cols)
        W = [(r >> lo) & ((1 << (hi - lo)) - 1) for r in self.A]
        piv, comb = [], {}                                  # comb[row] = mask over piv indices making up the reduced pivot row
        mult = [0] * self.rows
        alive = list(self.alive)
        for c in range(hi - lo):
            src = next((i for i in range(self.rows) if alive[i] and (W[i] >> c) & 1), None)
            if src is None:
                continue
            k = len(piv)
            piv.append((lo + c, src))
            alive[src] = False
            for i in range(self.rows):
                if i != src and (W[i] >> c) & 1:
                    W[i] ^= W[src]
                    mult[i] ^= mult[src] ^ (1 << k)         # rows (alive or pivots of this block) absorb pivot k
        p = self.payload
        p.zero_()
        p[0] = len(piv)
        for k, (c, r) in enumerate(piv):
            p[1 + 2 * k], p[2 + 2 * k] = c, r
        for i in range(self.rows):
            p[1 + 2 * BLOCK + i] = mult[i]
        return p

    def apply(self, b, payload):
        npiv = int(payload[0])
        piv = [(int(payload[1 + 2 * k]), int(payload[2 + 2 * k])) for k in range(npiv)]
        mult = [int(payload[1 + 2 * BLOCK + i]) for i in range(self.rows)]
        rowsnap = [self.A[r] for _, r in piv]               # the pivot rows as they are BEFORE the block (sources)
        for i in range(self.rows):
            m, acc = mult[i], 0
            while m:
                k = (m & -m).bit_length() - 1
                m &= m - 1
                acc ^= rowsnap[k]
            self.A[i] ^= acc
        for c, r in piv:
            self.alive[r] = False
            self.pivots.append((c, r))

    def finish_local(self):
        pass

    def tiles(self):
        t = torch.zeros(self.ntiles, self.rows, dtype=torch.int64)
        for ti in range(self.ntiles):
            for i in range(self.rows):
                t[ti, i] = (self.A[i] >> (TILE * ti)) & ((1 << TILE) - 1)
        self._t = t
        return t

    def solve(self):
        """rank 0, all tiles collected: the system is in reduced form (Gauss-Jordan per block, applied to every row)"""
        A = [sum(int(self._t[ti, i]) << (TILE * ti) for ti in range(self.ntiles)) for i in range(self.rows)]
        if any(self.alive[i] and (A[i] >> self.cols) & 1 for i in range(self.rows)):
            return None
        x = 0
        for c, r in self.pivots:
            x |= ((A[r] >> self.cols) & 1) << c
        return x, sorted(c for c, _ in self.pivots)


def _worker(rank, world, port, eqs, cols, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        eng = IntSlabEngine(eqs, cols, world, rank)
        res = slab.run_schedule(eng)
        if rank == 0:
            q.put(res)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _sharded(eqs, cols, world):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, eqs, cols, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return got


def test_two_rank_column_slab_schedule_matches_oracle():
    from oracle import gf2_oracle as O
    rng = random.Random(41)
    for rows, cols, cap, cons in ((70, 60, None, True), (90, 75, 50, True), (90, 75, 50, False)):
        eqs = random_system(rng, rows, cols, .5, cap, cons, 0)
        want = O.solve_words(O.eqs_to_aug(eqs, cols), rows, cols, 0)
        got = _sharded(eqs, cols, 2)
        if want["status"] != 0:
            assert got is None
            continue
        x, piv = got
        assert piv == list(want["pivcols"])
        assert x == O.words_to_int(want["origin"])          # free variables zero, pivot variables from the reduced rows
