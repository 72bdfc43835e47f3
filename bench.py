#!/usr/bin/env python3
"""bench.py -- headline benchmark: GF(2) row-XORs/s + solve wall-time, dense N x N solve_one.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--n 65536]

One "step" = one complete solve_one of a synthetic dense N x N GF(2) system that is already
resident in HBM (forward elimination + consistency check + back-substitution + export), through
the C ABI (gf2bv_solve_device).  Default workload = BASELINE.json configs[1]: 65536 x 65536,
full rank (seed committed below).  With --gpus N (launched by torch.distributed.run, one rank per
GPU) every rank solves its own independent systems -- the path shards by system, no data-path
collective -- and the solutions are gathered once at the end over RCCL ("scaling": "weak").

Rank 0 prints ONE JSON line (see the contract in the task statement), carrying
  roofline     : the sweep kernel (dominant): algorithmic bytes (16 B per active word per sweep)
                 / HIP-event time of the sweep launches, against 8 TB/s HBM
  cpu_baseline : the CPU oracle (M4RM-style "port", OpenMP) timed on this host on a bounded
                 sample of the same generator (rank 0, N=1 only)
torch is plumbing here: device memory, the RCCL gather, barriers.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402  (first, so its HIP runtime is the one loaded)
import torch.distributed as dist  # noqa: E402

from gf2bv_amd import hip  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
# synthetic-generator seeds whose N x N matrix has full rank (found with tools/find_full_rank_seed.py)
FULL_RANK_SEEDS = {65536: 1234, 32768: 1234}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", type=int, default=65536, help="system size N (rows = cols)")
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-n", type=int, default=65536, help="size of the bounded CPU-baseline sample")
    ap.add_argument("--batch-systems", type=int, default=16,
                    help="also time a gang batch of this many independent --batch-n systems (0 = skip); extra field, not `value`")
    ap.add_argument("--batch-n", type=int, default=32768)
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="do not bracket sweep launches with HIP events (roofline becomes null)")
    return ap.parse_args()


def cpu_baseline(n: int, seed: int) -> dict:
    """Time the CPU oracle (test infrastructure; here only as the reported baseline)."""
    from oracle import gf2_oracle as O
    L = O.lib()
    cores = L.gf2o_max_threads()

    def run(size, threads):
        L.gf2o_set_threads(threads)
        aug = O.gen_synthetic(size, size, seed)
        t0 = time.perf_counter()
        res = O.solve_words(aug, size, size, 0, algo=1)
        dt = time.perf_counter() - t0
        return res, dt, O.check_solution(aug, size, size, res["origin"])

    run(1024, cores)                                   # spin up the OpenMP team outside the timed region
    # the port does not scale to every core count (two barriers per panel): probe, then time the best
    probe = {}
    for t in sorted({1, 8, 16, 32, 64, cores}):
        if t <= cores:
            probe[t] = run(8192, t)[1]
    best = min(probe, key=probe.get)
    res, dt, bad = run(n, best)
    res1, dt1, _ = run(max(n // 2, 1024), 1)
    L.gf2o_set_threads(cores)
    panels = (n + 63) // 64
    return {
        "value": res["row_xors"] / dt, "unit": "row-XORs/s", "cores": best, "kind": "port",
        "sample": f"one {n}x{n} solve_one of the same synthetic generator (seed {seed}), "
                  f"oracle M4RM port (Gauss-Jordan, 8 byte-tables per 64-column panel, OpenMP over rows, "
                  f"best of {sorted(probe)} threads on a {cores}-thread host); libm4ri on this host: "
                  + str(__import__("ctypes.util").util.find_library("m4ri")),
        "seconds": dt, "rank": int(res["rank"]), "residual_rows": int(bad),
        "row_panels_per_s": n * panels / dt,          # table-count independent: (rows x 64-column panels) eliminated per second
        "single_core": {"value": res1["row_xors"] / dt1, "seconds": dt1, "n": max(n // 2, 1024)},
        "thread_probe_8192_seconds": {str(k): v for k, v in probe.items()},
        # true M4RI would be timed as well if the box had it; checked at run time
        "libm4ri_on_this_host": __import__("ctypes.util").util.find_library("m4ri"),
        "_origin": res["origin"], "_status": int(res["status"]),
    }


def batch_throughput(n: int, nsys: int, device: int) -> dict:
    """BASELINE configs[3] per-GPU share in miniature: nsys independent n x n systems resident in HBM, solved by
    gf2bv_solve_batch_device (lock-step gangs); every solution checked by the residual kernel."""
    stride = hip.padded_stride(n)
    buf = hip.DeviceBuffer(nsys * n * stride * 8, device)
    for i in range(nsys):
        hip.synth_device(buf.ptr + i * n * stride * 8, n, n, stride, 5000 + i, device=device)
    best, sols = None, None
    for _ in range(2):                                   # first pass warms the allocator
        t0 = time.perf_counter()
        sols = hip.solve_batch_device(buf.ptr, nsys, n * stride, n, n, stride, hip.MODE_SINGLE, device)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    bad = sum(hip.residual_device(buf.ptr + i * n * stride * 8, n, n, stride, s.origin, device=device)
              for i, s in enumerate(sols))
    xors = float(sum(s.stats["row_xors"] for s in sols))
    buf.free()
    return {"n": n, "systems": nsys, "ms_per_system": best / nsys * 1e3, "row_xors_per_s": xors / best,
            "residual_rows": int(bad), "all_solved": all(s.solved for s in sols)}


def pmc_traffic(n: int, g: int, t: int):
    """HBM bytes per bulk-update pass from the committed rocprofv3 PMC passes (FETCH_SIZE doubled per the
    gfx950 note in MI355X_MICROARCH.md, WRITE_SIZE as is); None when no profile of this config is committed."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None
    for rec in json.load(open(path)):
        if rec["n"] == n and rec["G"] == g and rec["T"] == t:
            return rec["hbm_bytes_per_pass"]
    return None


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (the product path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    n = args.n
    seed = args.seed if args.seed is not None else FULL_RANK_SEEDS.get(n, 1234)
    stride = hip.padded_stride(n)
    cw = (n + 63) // 64
    # the system lives in HBM before the timed region; the solver works on its own tile-major copy
    # (the row-major -> tile-major pass is part of every timed step), so one pristine matrix suffices
    mat = torch.empty(n * stride, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    hip.synth_device(mat.data_ptr(), n, n, stride, seed + rank, device=local_rank, stream=stream)
    torch.cuda.synchronize(dev)

    sols = torch.zeros(args.steps, cw, dtype=torch.int64, device=dev)
    stats = []

    def step(i: int):
        return hip.solve_device(mat.data_ptr(), n, n, stride, hip.MODE_SINGLE, device=local_rank,
                                stream=stream, time_kernels=not args.no_kernel_events)

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for k in range(args.steps):
        s = step(args.warmup + k)
        stats.append(s)
        sols[k].copy_(torch.from_numpy(s.origin.view(np.int64)))
    if world > 1:
        gathered = [torch.empty_like(sols) for _ in range(world)]
        dist.all_gather(gathered, sols)          # the single end-of-job gather (RCCL over xGMI)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # correctness gate on this rank: A x = b on the untouched input, by the independent residual kernel
    bad = hip.residual_device(mat.data_ptr(), n, n, stride, stats[-1].origin, device=local_rank, stream=stream)
    ok = torch.tensor([1 if (bad == 0 and all(s.solved for s in stats)) else 0], device=dev)
    row_xors_local = float(sum(s.stats["row_xors"] for s in stats))
    agg = torch.tensor([row_xors_local], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(agg)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if rank == 0:
        s0 = stats[-1].stats
        sweep_ms = float(np.mean([s.stats["ms_sweep"] for s in stats]))
        n_sweeps = s0["n_sweeps"]
        alg_bytes = 16.0 * s0["sweep_words"]
        roofline = None
        if sweep_ms > 0:
            achieved = alg_bytes / (sweep_ms * 1e-3) / 1e9
            g = s0["panels_per_sweep"]
            ceil = hip.stream_ceiling(2 << 30, local_rank)
            roofline = {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(n, g, s0["tables_per_sweep"] // g),
                "kernel": f"k_update<G={g},T={s0['tables_per_sweep'] // g}> (bulk update, {g} panels = {64 * g} pivots per pass)",
                "passes": n_sweeps,
                "alg_bytes_per_pass": alg_bytes / max(n_sweeps, 1),
                "avg_pass_ms": sweep_ms / max(n_sweeps, 1),
                # same-run practical ceilings of this device (plain streaming kernels, 2 GiB)
                "measured_rmw_stream_GBs": ceil["rmw_gbs"], "measured_read_stream_GBs": ceil["read_gbs"],
                "frac_of_measured_rmw": achieved / ceil["rmw_gbs"],
                # one pass applies G panels: HBM rate a one-panel-per-pass sweep would need for the same wall time
                "single_panel_equivalent_GBs": achieved * g,
            }
        out = {
            "metric": "GF(2) row-XORs/s (solve_one, dense NxN)", "value": float(agg.item()) / elapsed,
            "unit": "row-XORs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {
                "workload": f"synthetic dense {n}x{n} GF(2) solve_one, planted RHS, seed {seed}"
                            + (" (full rank)" if n in FULL_RANK_SEEDS and args.seed is None else ""),
                "systems_per_step_per_gpu": 1, "parallelism": f"independent systems x{world}",
                "tables_per_sweep": s0["tables_per_sweep"], "table_bits": s0["table_bits"],
                "tile_words": s0["tile_words"], "rank": int(stats[-1].rank),
            },
            "solve_wall_ms": {"eliminate": float(np.mean([s.stats["ms_eliminate"] for s in stats])),
                              "backsub": float(np.mean([s.stats["ms_backsub"] for s in stats])),
                              "export": float(np.mean([s.stats["ms_export"] for s in stats])),
                              "total_host": float(np.mean([s.stats["ms_total"] for s in stats]))},
            "parity_gate": {"residual_rows": int(bad), "all_ranks_ok": bool(ok.item())},
            # table-count independent work rate: (alive rows x 64-column panels) eliminated per second, whole job
            "row_panels_per_s": world * n * ((n + 63) // 64) / 2 / (elapsed / args.steps),
            "roofline": roofline,
        }
        if world == 1 and args.batch_systems > 0:
            out["batch_throughput"] = batch_throughput(args.batch_n, args.batch_systems, local_rank)
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(args.cpu_n, seed)
            # parity gate, second half: the same sample on the GPU must give the oracle's answer word for word
            m = args.cpu_n
            st2 = hip.padded_stride(m)
            buf = hip.DeviceBuffer(m * st2 * 8, local_rank)
            hip.synth_device(buf.ptr, m, m, st2, seed, device=local_rank)
            g = hip.solve_device(buf.ptr, m, m, st2, hip.MODE_SINGLE, device=local_rank)
            buf.free()
            o_status, o_origin = cb.pop("_status"), cb.pop("_origin")
            same = g.rank == cb["rank"] and g.status == o_status and np.array_equal(g.origin, o_origin)
            out["parity_gate"]["gpu_equals_cpu_oracle_on_sample"] = bool(same)
            out["cpu_baseline"] = cb
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
