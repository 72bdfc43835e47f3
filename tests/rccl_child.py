"""Body of the GPU suite's RCCL tests, run as a CHILD process (tests/child.py) so that a hang costs one test, not the suite.

    python tests/rccl_child.py control                       torch only: RCCL world-1 init + all_gather, the library never loaded
    python tests/rccl_child.py gather [--after-solve K] [--trim] [--lazy]
                                                             the end-of-job gather of gf2bv_amd.batch over backend "nccl";
                                                             --after-solve K: K x 32768^2 systems solved as gangs BEFORE the init
                                                             (round 5's driver run hung in exactly that order), --trim: both pools
                                                             given back + a device synchronisation in front of the init, --lazy: no
                                                             device_id (RCCL connects at the first collective instead of eagerly)
    python tests/rccl_child.py slab                          the column-slab schedule with the per-block broadcast over "nccl"
    python tests/rccl_child.py hang                          test-only: never returns (the runner must kill it)

Prints `CHILD_OK {...}` as its last line when everything it checked held."""
from __future__ import annotations

import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _rendezvous_env():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")


def control():
    import torch
    import torch.distributed as dist
    assert "gf2bv_amd" not in sys.modules
    _rendezvous_env()
    dev = torch.device("cuda", 0)
    t0 = time.perf_counter()
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    t_init = time.perf_counter() - t0
    x = torch.arange(8, device=dev)
    parts = [torch.empty_like(x)]
    dist.all_gather(parts, x)
    torch.cuda.synchronize()
    assert torch.equal(parts[0], x)
    dist.destroy_process_group()
    assert not any("gf2bv" in m for m in sys.modules)
    return {"mode": "control", "init_seconds": round(t_init, 3)}


def gather(argv):
    import numpy as np
    import torch
    import torch.distributed as dist

    from gf2bv_amd import batch, hip
    from oracle import gf2_oracle as O
    dev = torch.device("cuda", 0)
    info = {"mode": "gather", "argv": argv}
    if "--after-solve" in argv:
        k = int(argv[argv.index("--after-solve") + 1])
        n = 32768
        seeds = [5000 + i for i in range(k)]
        mats = batch.synth_shard(n, seeds, 0)
        recs, sols = batch.solve_shard(n, mats, 0)
        assert all(s.solved for s in sols) and all(s.stats["gang_systems"] >= 2 for s in sols)
        one = hip.solve_device(mats[0].data_ptr(), n, n, hip.padded_stride(n), 0)       # a single solve too: its probed stream pair
        assert one.solved and np.array_equal(one.origin, sols[0].origin)
        del mats, recs
        info["solved_before_init"] = k
        info["pool_idle_bytes_before_init"] = int(hip.pool_idle_bytes(0))
    if "--trim" in argv:
        hip.pool_trim(0)
        hip.host_pool_trim()
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
    _rendezvous_env()
    d = batch.init_process_group_guarded("nccl", None if "--lazy" in argv else dev, limit_s=100, rank=0, world_size=1)
    info["init_seconds"] = d["init_seconds"]
    try:
        n, total = 2048, 5
        lo, hi = batch.shard_bounds(total, 1, 0)
        seeds = [900 + i for i in range(lo, hi)]
        mats = batch.synth_shard(n, seeds, 0)
        recs, sols = batch.solve_shard(n, mats, 0)
        allrec = batch.gather_records(recs, total)
        torch.cuda.synchronize()
        assert allrec.is_cuda and torch.equal(allrec, recs)
        stride = hip.padded_stride(n)
        host = allrec.cpu().numpy()
        for i, (seed, s) in enumerate(zip(seeds, sols)):
            assert s.solved and s.rank >= n - 8
            assert hip.residual_device(mats[i].data_ptr(), n, n, stride, s.origin) == 0
            assert np.array_equal(host[i], batch.make_record(s.status, s.rank, s.origin))
        for i in (0, 4):                                                                 # word for word vs the oracle
            want = O.solve_words(O.gen_synthetic(n, n, seeds[i]), n, n, 0, algo=1)
            assert sols[i].status == want["status"] and sols[i].rank == want["rank"]
            assert np.array_equal(sols[i].pivots, want["pivcols"]) and np.array_equal(sols[i].origin, want["origin"])
        # an uneven split as rank 0 of a larger job would see it: padding rows never leak into the table
        part = batch.gather_records(recs[:3], 3)
        assert torch.equal(part, recs[:3])
    finally:
        dist.destroy_process_group()
    info["build_id"] = hip.build_id()
    return info


def slab_schedule():
    import numpy as np
    import torch
    import torch.distributed as dist

    from gf2bv_amd import batch, hip, slab
    from oracle import gf2_oracle as O
    dev = torch.device("cuda", 0)
    _rendezvous_env()
    d = batch.init_process_group_guarded("nccl", dev, limit_s=100, rank=0, world_size=1)
    try:
        for (n, seed) in ((4096, 21), (2500, 22)):
            stride = hip.padded_stride(n)
            aug = torch.empty(n * stride, dtype=torch.int64, device=dev)
            hip.synth_device(aug.data_ptr(), n, n, stride, seed, device=0, stream=torch.cuda.current_stream(dev).cuda_stream)
            torch.cuda.synchronize(dev)
            sol = slab.solve_one_sharded(aug, n, n, stride, 0, always_broadcast=True)
            want = O.solve_words(O.gen_synthetic(n, n, seed), n, n, 0, algo=1)
            assert sol.status == want["status"] and sol.rank == want["rank"]
            assert np.array_equal(sol.pivots, want["pivcols"]) and np.array_equal(sol.origin, want["origin"])
    finally:
        dist.destroy_process_group()
    return {"mode": "slab", "init_seconds": d["init_seconds"]}


def main():
    mode = sys.argv[1]
    if mode == "hang":
        print("hanging on purpose", flush=True)
        while True:
            time.sleep(3600)
    out = {"control": control, "gather": lambda: gather(sys.argv[2:]), "slab": slab_schedule}[mode]()
    print("CHILD_OK " + json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
