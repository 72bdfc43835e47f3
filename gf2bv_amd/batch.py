"""Batch sharding of independent systems across the GPUs of a node.

The reference solves one system per ``m4ri_solve`` call and has no distributed code
(SURVEY.md 2.1 / 8e); independent systems (one per output bit / per instance in the recovery
examples) are the natural shard unit.  One process per GPU, contiguous blocks of systems per
rank, NO collective on the data path; a single all_gather of fixed-size solution records at
the end (RCCL over xGMI with backend "nccl", gloo on CPU for tests).

Record layout (int64 words): [status, rank, origin word 0 .. origin word cw-1].
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(nsys: int, world: int, rank: int) -> tuple:
    """Contiguous block [lo, hi) of systems owned by `rank`; the first nsys % world ranks get one more."""
    base, extra = divmod(nsys, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def record_words(cols: int) -> int:
    return 2 + (cols + 63) // 64


def make_record(status: int, rank: int, origin_words: np.ndarray) -> np.ndarray:
    rec = np.empty(2 + len(origin_words), dtype=np.int64)
    rec[0], rec[1] = status, rank
    rec[2:] = np.asarray(origin_words, dtype=np.uint64).view(np.int64)
    return rec


def gather_records(local: torch.Tensor, nsys: int, group=None) -> torch.Tensor:
    """local: [hi-lo, R] int64 records of this rank's block -> [nsys, R] on every rank.

    Pads every block to the largest block so a single all_gather suffices."""
    if not (dist.is_available() and dist.is_initialized()):
        return local
    world = dist.get_world_size(group)
    width = local.shape[1]
    biggest = (nsys + world - 1) // world
    padded = torch.zeros(biggest, width, dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    if padded.is_cuda and dist.get_backend(group) == "gloo":
        # gloo moves host memory only (tests: several ranks sharing one GPU); RCCL ("nccl") gathers device tensors directly
        host = padded.cpu()
        hparts = [torch.empty_like(host) for _ in range(world)]
        dist.all_gather(hparts, host, group=group)
        parts = [h.to(padded.device) for h in hparts]
    else:
        parts = [torch.empty_like(padded) for _ in range(world)]
        dist.all_gather(parts, padded, group=group)
    out = torch.empty(nsys, width, dtype=local.dtype, device=local.device)
    for r in range(world):
        lo, hi = shard_bounds(nsys, world, r)
        out[lo:hi] = parts[r][: hi - lo]
    return out


def rank_diagnostics(device_index: int) -> dict:
    """What a record needs to tell one rank's GPU from another's: device, PCI bus id, name, RCCL version.  Touches the device
    properties only (no allocation, no stream)."""
    info = {"rank": int(os.environ.get("RANK", "0")), "local_rank": int(os.environ.get("LOCAL_RANK", "0")),
            "world": int(os.environ.get("WORLD_SIZE", "1")), "device": device_index, "pid": os.getpid()}
    try:
        p = torch.cuda.get_device_properties(device_index)
        info.update(name=p.name, hbm_gib=round(p.total_memory / 2 ** 30, 1), pci_bus_id=getattr(p, "pci_bus_id", None),
                    pci_device_id=getattr(p, "pci_device_id", None), visible=os.environ.get("HIP_VISIBLE_DEVICES"))
    except Exception as exc:                                # pragma: no cover - diagnostics must never be what fails
        info["device_error"] = repr(exc)
    try:
        info["rccl"] = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception as exc:                                # pragma: no cover
        info["rccl"] = repr(exc)
    return info


def init_process_group_guarded(backend: str, device: torch.device | None = None, limit_s: float = 180.0, **kw) -> dict:
    """`dist.init_process_group` that cannot hang the job: a watchdog thread ends the PROCESS with exit code 3 when the call has not
    returned after limit_s, after printing this rank's diagnostics, every thread's Python stack and -- if NCCL_DEBUG_FILE is set --
    the tail of the RCCL log to stderr.  (Round 5: the driver's GPU run waited 240 s inside RCCL's eager single-device connect and
    could not say where.)  Prints one JSON diagnostic line per rank to stderr BEFORE the call; returns it with `init_seconds`."""
    import faulthandler
    import json
    import sys
    import threading
    import time

    idx = device.index if device is not None and device.index is not None else 0
    info = rank_diagnostics(idx) if backend == "nccl" else {"rank": int(os.environ.get("RANK", "0")), "pid": os.getpid()}
    info["backend"] = backend
    print("[gf2bv rank] " + json.dumps(info), file=sys.stderr, flush=True)
    done = threading.Event()

    def watchdog():
        if done.wait(limit_s):
            return
        print(f"[gf2bv rank] init_process_group({backend!r}) did not return within {limit_s:.0f} s -- giving up: "
              + json.dumps(info), file=sys.stderr, flush=True)
        faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
        path = os.environ.get("NCCL_DEBUG_FILE")
        if path:
            try:
                sys.stderr.write("--- RCCL log (tail) ---\n" + open(path, errors="replace").read()[-4000:] + "\n")
            except OSError:
                pass
        sys.stderr.flush()
        os._exit(3)

    threading.Thread(target=watchdog, name="gf2bv-init-watchdog", daemon=True).start()
    t0 = time.perf_counter()
    try:
        if backend == "nccl" and device is not None:
            dist.init_process_group(backend, device_id=device, **kw)
        else:
            dist.init_process_group(backend, **kw)
    finally:
        done.set()
    info["init_seconds"] = round(time.perf_counter() - t0, 3)
    print(f"[gf2bv rank] process group up after {info['init_seconds']} s", file=sys.stderr, flush=True)
    return info


def rank_plan(total: int, n: int, world: int, rank: int, hbm_bytes: int = 288 * 10 ** 9) -> dict:
    """What rank `rank` of `world` does in the batch job of `total` independent n x n systems -- WITHOUT touching a GPU (bench.py
    --dry-run-ranks; the gang size comes from the library's own planner, gf2bv_plan_gang): its block of systems, the gangs it runs
    them in, where its records land in the gathered table, and what it keeps resident in HBM."""
    from . import hip
    lo, hi = shard_bounds(total, world, rank)
    nsys = hi - lo
    stride = hip.padded_stride(n)
    in_bytes = nsys * n * stride * 8
    free_b = max(0, hbm_bytes - in_bytes)
    gang = int(hip.lib().gf2bv_plan_gang(nsys, n, n, free_b)) if nsys else 0
    gangs = [(lo + g0, min(hi, lo + g0 + gang)) for g0 in range(0, nsys, gang)] if gang else []
    work = gang * n * stride * 8                       # one gang's tile-major working copies (the library pools up to six such buffers)
    rw = record_words(n)
    return {"rank": rank, "device": rank, "systems": [lo, hi], "gang_size": gang, "gangs": gangs, "host_threads": min(2, max(1, len(gangs))),
            "record_words": rw, "record_rows_in_gathered_table": [lo, hi], "record_bytes": nsys * rw * 8,
            "all_gather_rows_per_rank": (total + world - 1) // world,
            "resident_bytes": {"inputs": in_bytes, "working_per_gang": work, "pool_cap": min(48 << 30, hbm_bytes // 6),
                               "peak_estimate": in_bytes + 3 * work + 3 * (work // 8)}, "hbm_bytes": hbm_bytes}


def synth_shard(n: int, seeds: list, device_index: int, mats: torch.Tensor | None = None) -> torch.Tensor:
    """This rank's block of synthetic n x n systems, generated in HBM on torch's current stream (asynchronous)."""
    from . import hip
    stride = hip.padded_stride(n)
    dev = torch.device("cuda", device_index)
    if mats is None:
        mats = torch.empty(len(seeds), n * stride, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    for i, seed in enumerate(seeds):
        hip.synth_device(mats[i].data_ptr(), n, n, stride, seed, device=device_index, stream=stream)
    return mats


def solve_shard(n: int, mats: torch.Tensor, device_index: int, time_kernels: bool = False):
    """Solve this rank's block (mats: [nsys, n * stride] int64, resident in HBM) on its GPU as lock-step gangs.

    Returns (records [nsys, R] int64 on the GPU, list of hip.Solution).  The gangs are ordered after torch's
    current stream -- the stream that produced `mats` (gf2bv_solve_batch_device's ordering contract)."""
    from . import hip
    stride = hip.padded_stride(n)
    dev = torch.device("cuda", device_index)
    nsys = int(mats.shape[0])
    stream = torch.cuda.current_stream(dev).cuda_stream
    sols = hip.solve_batch_device(mats.data_ptr(), nsys, n * stride, n, n, stride, hip.MODE_SINGLE,
                                  device=device_index, stream=stream, time_kernels=time_kernels) if nsys else []
    recs = np.stack([make_record(s.status, s.rank, s.origin) for s in sols]) if sols else \
        np.zeros((0, record_words(n)), dtype=np.int64)
    return torch.from_numpy(recs).to(dev), sols


def solve_synthetic_shard(n: int, seeds: list, device_index: int, mats: torch.Tensor | None = None) -> torch.Tensor:
    """Generate and solve this rank's block of synthetic n x n systems; returns [len(seeds), R] records (on GPU)."""
    mats = synth_shard(n, seeds, device_index, mats)
    return solve_shard(n, mats, device_index)[0]
