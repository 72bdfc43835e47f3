"""ONE system over the GPUs of a node: column-slab sharding.

The reference factorises one matrix with one ``_mzd_pluq`` call (gf2bv/_internal.c:431-433).  Here rank r of
``world`` processes (one per GPU) owns the column tiles t with t % world == r; per block of 4 panels the owner of the
block's window runs the panel path and broadcasts the block's records -- pivots, source rows, combinations and rows x
32 bytes of per-row multipliers -- and every rank applies the block to the tiles it owns (k_block_trsm + k_update of
libgf2bv_hip.so).  ONE broadcast per block on the data path (RCCL over xGMI with backend "nccl"); after the last block
the tiles are collected on rank 0, which finishes like a single-GPU solve.  The schedule is the C ABI's
(include/gf2bv_hip.h, gf2bv_slab_*); this module supplies the collectives through torch.distributed and nothing else.

The engine behind the schedule is an object with the methods of ``HipSlabEngine``; the CPU-side protocol test
(tests/test_slab_protocol_cpu.py) plugs in a small integer engine instead.
"""
from __future__ import annotations

import ctypes

import torch
import torch.distributed as dist


class HipSlabEngine:
    """The gf2bv_slab_* entry points for one rank.  `aug` = the full row-major system on this rank's GPU."""

    def __init__(self, aug_ptr: int, rows: int, cols: int, stride: int, world: int, rank: int, device_index: int):
        from . import hip
        self.hip, self.L = hip, hip.lib()
        self.rows, self.cols, self.world, self.rank = rows, cols, world, rank
        self.dev = torch.device("cuda", device_index)
        self.ntiles = int(self.L.gf2bv_slab_tiles(cols))
        words = int(self.L.gf2bv_slab_work_words(rows, cols))
        self.work = torch.empty(words, dtype=torch.int64, device=self.dev)          # tile-major working matrix
        h = ctypes.c_void_p()
        hip._check(self.L.gf2bv_slab_open(aug_ptr, rows, cols, stride, self.work.data_ptr(), words, world, rank,
                                          device_index, ctypes.byref(h)))
        self.h = h
        self.nblocks = int(self.L.gf2bv_slab_blocks(h))
        # two payload buffers: block b uses buffer b & 1 on every rank, so the export of block b + 1 never waits for the
        # broadcast of block b (gf2bv_slab_factor_on)
        nw = int(self.L.gf2bv_slab_payload_bytes(h)) // 8
        self.payloads = [torch.zeros(nw, dtype=torch.int64, device=self.dev) for _ in range(2)]

    def owner(self, b: int) -> int:
        return int(self.L.gf2bv_slab_owner(self.h, b))

    def _stream(self):
        # the stream torch.distributed enqueues its collectives behind: the library orders its own streams against it
        # with events (gf2bv_slab_factor_on / _apply_on), no call waits for the device
        return torch.cuda.current_stream(self.dev).cuda_stream or None

    def recv_buffer(self, b: int) -> torch.Tensor:
        return self.payloads[b & 1]

    def factor(self, b: int) -> torch.Tensor:
        buf = self.payloads[b & 1]
        self.hip._check(self.L.gf2bv_slab_factor_on(self.h, b, buf.data_ptr(), self._stream()))
        return buf

    def apply(self, b: int, payload: torch.Tensor):
        self.hip._check(self.L.gf2bv_slab_apply_on(self.h, b, payload.data_ptr(), self._stream()))

    def finish_local(self):
        self.hip._check(self.L.gf2bv_slab_finish_local(self.h))

    def tiles(self) -> torch.Tensor:
        """[ntiles, slab_words] view of the working matrix; rank r owns rows r::world."""
        return self.work.view(self.ntiles, -1)

    def solve(self):
        h = ctypes.c_void_p()
        self.hip._check(self.L.gf2bv_slab_solve(self.h, ctypes.byref(h)))
        return self.hip._take(h, self.hip.MODE_SINGLE)

    def close(self):
        if self.h:
            self.L.gf2bv_slab_close(self.h)
            self.h = None


def _broadcast(t: torch.Tensor, src: int, group=None):
    """dist.broadcast; staged through the host when the backend cannot move device tensors itself (gloo in tests)."""
    if t.is_cuda and dist.get_backend(group) == "gloo":
        c = t.cpu()
        dist.broadcast(c, src=src, group=group)
        if dist.get_rank(group) != src:
            t.copy_(c)
    else:
        dist.broadcast(t, src=src, group=group)


def run_schedule(engine, group=None, always_broadcast: bool = False):
    """The column-slab schedule on an initialised process group; returns rank 0's result (None elsewhere).
    `always_broadcast`: issue the per-block collective at world size 1 as well (tests: the RCCL path on a one-GPU box)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    for b in range(engine.nblocks):
        src = engine.owner(b)
        payload = engine.factor(b) if src == rank else (engine.recv_buffer(b) if hasattr(engine, "recv_buffer") else engine.payload)
        if world > 1 or always_broadcast:
            _broadcast(payload, src, group)                 # the ONE collective of the block (stream-ordered on the GPU)
        engine.apply(b, payload)
    engine.finish_local()
    if world > 1:
        tiles = engine.tiles()
        for r in range(world):                              # every rank's tiles, collected (rank 0 is the one that needs them)
            part = tiles[r::world].contiguous()
            _broadcast(part, r, group)
            if r != rank and rank == 0:
                tiles[r::world] = part
        if tiles.is_cuda:
            torch.cuda.synchronize(tiles.device)
    return engine.solve() if rank == 0 else None


def solve_one_sharded(aug: torch.Tensor, rows: int, cols: int, stride: int, device_index: int, group=None,
                      always_broadcast: bool = False):
    """solve_one of the system `aug` (row-major augmented words, int64 tensor on this rank's GPU, the same on every
    rank) with its columns sharded over the ranks of `group`.  rank 0 returns a hip.Solution, the others None."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    eng = HipSlabEngine(aug.data_ptr(), rows, cols, stride, world, rank, device_index)
    try:
        return run_schedule(eng, group, always_broadcast)
    finally:
        eng.close()
