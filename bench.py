#!/usr/bin/env python3
"""bench.py -- headline benchmark: GF(2) row-XORs/s + solve wall-time, dense N x N solve_one.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload single|batch] [--n 65536]

Workloads, all through the C ABI, inputs resident in HBM before the timed region:

  single (the default at EVERY N; BASELINE.json configs[1]): one "step" = one complete solve_one of a synthetic dense
      65536 x 65536 full-rank system PER GPU (tile-major copy + forward elimination + consistency check +
      back-substitution + export) by gf2bv_solve_device; at N > 1 every rank solves its own system (generator seed +
      rank), one all_gather of the solutions at the end -- per-GPU work is fixed, "scaling" is "weak", and the lines of
      N = 1, 2, 4, 8 are the SAME workload, so the driver's scaling curve compares like with like.
      Every default line also carries `batch_c4`: BASELINE.json configs[3], the job of 512 independent 32768 x 32768
      systems sharded in contiguous blocks over the N ranks (1 warm-up + 1 timed step; at N = 1 all 512 on the one GPU,
      73 GB resident -- the anchor of the strong-scaling curve `batch_c4.systems_per_s` over N), with its own roofline
      block.  The N = 1 line adds `target_262144` (the north-star size, 2 steps after 1 warm-up) and `cpu_baseline`.
      None of these extras is ever `value`.
  batch  (`--workload batch`; BASELINE.json configs[3] as the line's own `value`): 512 independent 32768 x 32768
      systems, sharded in contiguous blocks over the ranks (gf2bv_amd.batch.shard_bounds), every rank solving its
      block with gf2bv_solve_batch_device (lock-step gangs) -- no collective on the data path -- then ONE all_gather of
      the fixed-size records [status, rank, origin] (RCCL over xGMI).  One "step" = the whole 512-system job; the total
      is fixed, so "scaling" is "strong".

Launched for N > 1 as `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (one rank per GPU).
Rank 0 prints ONE JSON line carrying
  roofline     : the bulk-update kernel (dominant): algorithmic bytes (16 B per active word per pass) / HIP-event time
                 of its launches, against 8 TB/s HBM
  cpu_baseline : the CPU oracle (M4RM-style "port", OpenMP) timed on this host on a bounded sample (rank 0, N = 1 only)
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

from gf2bv_amd import batch, hip  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
# synthetic-generator seeds whose N x N matrix has full rank (found with tools/find_full_rank_seed.py)
FULL_RANK_SEEDS = {65536: 1234, 32768: 1234, 262144: 1242}
LDS_BYTES_PER_CLK_CU = 256.0   # ds_read_b128, conflict-free (MI355X_MICROARCH.md, LDS table)
BATCH_SEED0 = 5000             # system i of the batch workload uses generator seed BATCH_SEED0 + i


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=["auto", "single", "batch", "sharded"], default="auto",
                    help="auto = single at every N (configs[1] per GPU, weak scaling) with the configs[3] job as the extra block "
                         "`batch_c4`; batch = configs[3] as the line itself; sharded = ONE --n system with its columns over "
                         "the ranks (SURVEY 8f-1, strong scaling)")
    ap.add_argument("--n", "--size", dest="n", type=int, default=65536,
                    help="single: system size N (rows = cols); spell it --size under torch.distributed.run, whose own parser "
                         "takes a bare --n for an abbreviation of its options")
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--no-plain-leg", action="store_true", help="skip the repeat of the timed steps under GF2BV_PLAIN=1 (profiling runs: keeps the kernel statistics to the default path)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-n", type=int, default=None, help="size of the bounded CPU-baseline sample (default: 65536 single / 32768 batch)")
    ap.add_argument("--batch-total", type=int, default=512, help="batch: systems in the whole job (sharded over the ranks)")
    ap.add_argument("--batch-n", type=int, default=32768)
    ap.add_argument("--no-batch-c4", action="store_true",
                    help="single: skip the `batch_c4` block (the configs[3] job of --batch-total x --batch-n systems over the ranks)")
    ap.add_argument("--target-n", type=int, default=262144,
                    help="single: also run the north-star size (1 warm-up + 2 steps) as `target_262144` (0 = skip)")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="single, N = 1: skip c3_mt19937 (configs[2]), c5_xoshiro (configs[4]) and the host-resident 65536^2 solve")
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="do not bracket bulk-update launches with HIP events (roofline becomes null)")
    ap.add_argument("--dry-run-ranks", action="store_true",
                    help="print every rank's plan of the multi-GPU batch job (shard bounds, gangs, record offsets, resident bytes) and "
                         "exit WITHOUT solving or touching a GPU: runs under torch.distributed.run with gloo on a CPU box")
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
        # (round 5) oracle/m4ri_timing.c: when a libm4ri.so can be opened, mzd_pluq + mzd_pluq_solve_left are timed on an n x n
        # mzd_randomize matrix (`kind` stays "port" for `value`; this block is the real library beside it)
        "m4ri": O.m4ri_time(n),
        "_origin": res["origin"], "_status": int(res["status"]),
    }


def cpu_baseline_list(eqs, cols: int, mode: int, reps: int) -> dict:
    """The CPU oracle on a list-of-int system (the boundary the reference's m4ri_solve takes): median wall time of `reps`
    calls and the result -- the timed CPU baseline beside the c3 / c5 legs, and their checker."""
    from oracle import gf2_oracle as O
    ts, res = [], None
    for _ in range(reps):
        t0 = time.perf_counter()
        res = O.m4ri_solve(list(eqs), cols, mode)
        ts.append(time.perf_counter() - t0)
    return {"ms_median": sorted(ts)[len(ts) // 2] * 1e3, "result": res, "threads": int(O.lib().gf2o_max_threads())}


def gpu_equals_oracle(cb: dict, m: int, seed: int, device: int) -> bool:
    """parity gate, second half: the CPU sample solved on the GPU must give the oracle's answer word for word"""
    st2 = hip.padded_stride(m)
    buf = hip.DeviceBuffer(m * st2 * 8, device)
    hip.synth_device(buf.ptr, m, m, st2, seed, device=device)
    g = hip.solve_device(buf.ptr, m, m, st2, hip.MODE_SINGLE, device=device)
    buf.free()
    o_status, o_origin = cb.pop("_status"), cb.pop("_origin")
    return bool(g.rank == cb["rank"] and g.status == o_status and np.array_equal(g.origin, o_origin))


def pmc_traffic(n: int, g: int, t: int, kernel: str = "k_update16", gang: int = 0):
    """HBM bytes per launch of the dominant bulk kernel from the committed rocprofv3 PMC passes (FETCH_SIZE doubled per the
    gfx950 note in MI355X_MICROARCH.md, WRITE_SIZE as is); None when no profile of this config is committed.
    gang > 0: a launch of the bulk update of a gang of that many n x n systems (the PMC pass measured one gang; per system and launch)."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None
    for rec in json.load(open(path)):
        if (rec["n"] == n and rec["G"] == g and rec["T"] == t and rec.get("current", True)
                and rec.get("kernel", "k_update16") == kernel and bool(rec.get("gang")) == (gang > 0)):
            return rec["hbm_bytes_per_pass_per_system"] * gang if gang else rec["hbm_bytes_per_pass"]
    return None


_LDS_CLOCK = {}


def lds_clock(device: int) -> dict:
    """shader clock under an LDS-bound load + CU count of this device, measured once per process"""
    if device not in _LDS_CLOCK:
        c = hip.lds_clock(device)
        c["cus"] = torch.cuda.get_device_properties(device).multi_processor_count
        c["lds_peak_GBs"] = c["cus"] * LDS_BYTES_PER_CLK_CU * c["shader_mhz"] * 1e6 / 1e9
        _LDS_CLOCK[device] = c
    return _LDS_CLOCK[device]


def roofline_block(stats_list, sweep_ms_total: float, n: int, device: int, ceil: dict | None, note: str | None = None):
    """`stats_list`: the gf2bv_stats of the solves whose bulk-update launches took `sweep_ms_total` (HIP events)."""
    if sweep_ms_total <= 0:
        return None
    s0 = stats_list[0]
    alg_bytes = 16.0 * float(sum(s["sweep_words"] for s in stats_list))
    launches = sum(s["n_sweeps"] for s in stats_list)
    achieved = alg_bytes / (sweep_ms_total * 1e-3) / 1e9
    g = s0["panels_per_sweep"]
    t = s0["tables_per_sweep"] // g
    # What bounds the dominant kernel.  One block per trip through HBM (k_update16): the memory side -- "hbm".  Outer passes
    # of the two-level elimination (k_update16k) apply K blocks per trip: HBM moves 1/K of the sweep-word bytes and the
    # kernel is bound by its table lookups in the LDS -- "lds": every 64-bit sweep-word costs g x t lookups of 16 B per
    # 16-byte row segment = 8 x g x t bytes of ds_read_b128 traffic (256 B for g = 4, t = 8); `lds` prices that against
    # CUs x 256 B/clk x the shader clock measured under an LDS-bound load in this run.  `frac` stays the HBM-unit figure
    #  in both cases, so lines of all rounds compare.
    outer_dominant = 2 * s0.get("outer_blocks", 0) > s0["n_sweeps"]
    lds = None
    if outer_dominant:
        clk = lds_clock(device)
        lds_bytes = 8.0 * g * t * float(sum(s["sweep_words"] for s in stats_list))
        lds_rate = lds_bytes / (sweep_ms_total * 1e-3) / 1e9
        lds = {"lookup_bytes_total": lds_bytes, "achieved_GBs": lds_rate, "peak_GBs": clk["lds_peak_GBs"],
               "frac": lds_rate / clk["lds_peak_GBs"], "shader_mhz_under_lds_load": clk["shader_mhz"], "cus": clk["cus"],
               "probe_bytes_per_clk_cu": clk["lds_bytes_per_clk_cu"],
               "note": "table lookups only (the table builds, ~9 % more LDS traffic, are not counted); kernel time = the time during "
                       "which at least one bulk-update launch runs (launches side by side -- the two halves of an outer pass on "
                       "their two streams, the next panel's inner updates -- count once)"}
    out = {
        "bound": "lds" if outer_dominant else "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBS, "lds": lds,
        # PMC bytes per launch of the kernel that dominates this solve (k_update16k where outer passes cover most blocks)
        "traffic": pmc_traffic(n, g, t, "k_update16k" if 2 * s0.get("outer_blocks", 0) > s0["n_sweeps"] else "k_update16"),
        "kernel": (f"k_update16 (bulk update on 16-byte tiles, {g} panels = {64 * g} pivots = {g * t} byte-field tables per pass)"
                   + (f" + k_update16k (outer passes of the two-level elimination: {s0['outer_blocks']} of the {s0['n_sweeps']} blocks, "
                      f"several blocks per trip through HBM)" if s0.get("outer_blocks") else "")),
        "unit_of_work": "sweep-word = one 64-bit word taking one block of 256 pivots = 16 B (SURVEY 8d)",
        "alg_bytes_total": alg_bytes, "kernel_ms_total": sweep_ms_total,
        # what the launches moved through HBM (read + written): below the algorithmic bytes where outer passes of the
        # two-level elimination apply several blocks per trip
        "hbm_bytes_total": 16.0 * float(sum(s.get("hbm_words", s["sweep_words"]) for s in stats_list)),
        "bulk_launches": int(sum(s.get("bulk_launches", s["n_sweeps"]) for s in stats_list)),
        "outer_blocks_per_solve": int(s0.get("outer_blocks", 0)),
        # one pass applies G panels: HBM rate a one-panel-per-pass sweep would need for the same wall time
        "single_panel_equivalent_GBs": achieved * g,
    }
    if ceil:
        # same-run context, not a ceiling: this library's own persistent-workgroup streams (profiles/r03_stream_ceiling.txt has
        # the independent references: an in-place stream reaches 6.4 TB/s = 0.80 of spec as one element per thread)
        out.update({"persistent_rmw_stream_GBs": ceil["rmw_gbs"], "persistent_read_stream_GBs": ceil["read_gbs"]})
    if note:
        out["note"] = note
    return out, launches


def finish_roofline(roofline: dict, s0: dict, blocks: int):
    """per-block and per-launch figures of a roofline block (`blocks` = 256-pivot blocks applied over the timed solves)"""
    nl = max(roofline["bulk_launches"], 1)
    roofline["passes"] = s0["n_sweeps"]                   # blocks of 256 pivots per solve
    roofline["alg_bytes_per_pass"] = roofline["alg_bytes_total"] / max(blocks, 1)
    roofline["avg_pass_ms"] = roofline["kernel_ms_total"] / max(blocks, 1)
    roofline["alg_bytes_per_launch"] = roofline["alg_bytes_total"] / nl
    roofline["hbm_bytes_per_launch"] = roofline["hbm_bytes_total"] / nl
    roofline["avg_launch_ms"] = roofline["kernel_ms_total"] / nl
    # (kernel_ms_total = gf2bv_stats.ms_sweep: launches of one stream add up -- every one-level plan, the headline -- while launches
    # that run side by side count once: since late round 5 an outer pass of the two-level elimination is two launches on two
    # streams and the next panel's inner updates run beside them; avg_launch_ms is then the time a launch ACCOUNTS for, not its
    # own duration.  elimination_* below is the whole forward elimination, panel path included)


def timed_single(mat: torch.Tensor, n: int, stride: int, steps: int, warmup: int, device: int, dev, world: int,
                 kernel_events: bool):
    """W untimed + K timed solve_one steps of the resident system; returns (elapsed seconds over ranks, stats, sols)."""
    stream = torch.cuda.current_stream(dev).cuda_stream
    cw = (n + 63) // 64
    sols = torch.zeros(steps, cw, dtype=torch.int64, device=dev)
    stats = []

    def step():
        return hip.solve_device(mat.data_ptr(), n, n, stride, hip.MODE_SINGLE, device=device, stream=stream,
                                time_kernels=kernel_events)

    for _ in range(warmup):
        step()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for k in range(steps):
        s = step()
        stats.append(s)
        sols[k].copy_(torch.from_numpy(s.origin.view(np.int64)))
    if world > 1:                                # the single end-of-job gather (RCCL over xGMI)
        if dist.get_backend() == "gloo":         # (tests: ranks sharing one GPU; gloo gathers host memory only)
            host = sols.cpu()
            dist.all_gather([torch.empty_like(host) for _ in range(world)], host)
        else:
            dist.all_gather([torch.empty_like(sols) for _ in range(world)], sols)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed, stats, sols


def run_single(args, world, rank, local_rank, dev):
    n = args.n
    seed = args.seed if args.seed is not None else FULL_RANK_SEEDS.get(n, 1234)
    stride = hip.padded_stride(n)
    # the system lives in HBM before the timed region; the solver works on its own tile-major copy
    # (the row-major -> tile-major pass is part of every timed step), so one pristine matrix suffices
    mat = torch.empty(n * stride, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    hip.synth_device(mat.data_ptr(), n, n, stride, seed + rank, device=local_rank, stream=stream)
    torch.cuda.synchronize(dev)
    elapsed, stats, _ = timed_single(mat, n, stride, args.steps, args.warmup, local_rank, dev, world,
                                     not args.no_kernel_events)

    # correctness gate on this rank: A x = b on the untouched input, by the independent residual kernel
    bad = hip.residual_device(mat.data_ptr(), n, n, stride, stats[-1].origin, device=local_rank, stream=stream)
    # (round 6) the same steps with GF2BV_PLAIN=1 -- no stream-pair probes, no XCD pinning, events instead of memory gates, no
    # optimistic enqueue -- on rank 0 at N = 1: what a box costs where those heuristics are off; never part of `value`
    plain = None
    if world == 1 and os.environ.get("GF2BV_PLAIN") is None and not args.no_plain_leg:
        os.environ["GF2BV_PLAIN"] = "1"
        try:
            p_elapsed, p_stats, _ = timed_single(mat, n, stride, args.steps, 1, local_rank, dev, world, False)
            plain = {"ms_per_step": p_elapsed / args.steps * 1e3,
                     "same_answer": bool(all(np.array_equal(a.origin, b.origin) and a.rank == b.rank for a, b in zip(stats, p_stats)))}
        finally:
            del os.environ["GF2BV_PLAIN"]
    ok = torch.tensor([1 if (bad == 0 and all(s.solved for s in stats)) else 0], device=dev)
    agg = torch.tensor([float(sum(s.stats["row_xors"] for s in stats))], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(agg)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    del mat
    # the north-star leg BEFORE the 73 GB batch job: behind it the 8 GiB working copy comes out of a device heap that has just
    # held and released hundreds of buffers, and the same solve reads ~3 % slower (1.308 against 1.271 s on one box)
    ceil = hip.stream_ceiling(2 << 30, local_rank) if rank == 0 else None
    target = None
    if world == 1 and args.target_n > 0 and n == 65536:
        torch.cuda.empty_cache()
        target = target_leg(args.target_n, local_rank, dev, ceil)
    c4 = None
    if not args.no_batch_c4 and args.batch_total > 0:
        torch.cuda.empty_cache()
        c4 = batch_job(args, world, rank, local_rank, dev, steps=1, warmup=1)       # every rank takes part (barriers, gather)
    if rank != 0:
        return None
    s0 = stats[-1].stats
    rl = roofline_block([s.stats for s in stats], float(sum(s.stats["ms_sweep"] for s in stats)), n, local_rank, ceil)
    roofline = None
    if rl:
        roofline, launches = rl
        finish_roofline(roofline, s0, launches)
        elim_ms = float(sum(s.stats["ms_eliminate"] for s in stats))
        roofline["elimination_GBs"] = roofline["alg_bytes_total"] / (elim_ms * 1e-3) / 1e9
        roofline["elimination_frac"] = roofline["elimination_GBs"] / HBM_PEAK_GBS
        roofline["hbm_real_frac"] = roofline["hbm_bytes_total"] / (elim_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
    out = {
        "metric": "GF(2) row-XORs/s (solve_one, dense NxN)", "value": float(agg.item()) / elapsed,
        "unit": "row-XORs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {
            "workload": f"synthetic dense {n}x{n} GF(2) solve_one, planted RHS, seed {seed}"
                        + (" (full rank)" if n in FULL_RANK_SEEDS and args.seed is None else ""),
            "systems_per_step_per_gpu": 1,
            "parallelism": f"independent systems x{world}" + (", one all_gather of the solutions at the end" if world > 1 else ""),
            "tables_per_sweep": s0["tables_per_sweep"], "table_bits": s0["table_bits"],
            "tile_words": s0["tile_words"], "rank": int(stats[-1].rank),
        },
        "solve_wall_ms": {"eliminate": float(np.mean([s.stats["ms_eliminate"] for s in stats])),
                          "backsub": float(np.mean([s.stats["ms_backsub"] for s in stats])),
                          "export": float(np.mean([s.stats["ms_export"] for s in stats])),
                          "total_host": float(np.mean([s.stats["ms_total"] for s in stats]))},
        "parity_gate": {"residual_rows": int(bad), "all_ranks_ok": bool(ok.item()),
                        "x_equals_planted": (bool(all(np.array_equal(s.origin, hip.planted_solution(n, seed + rank)) for s in stats))
                                             if stats[-1].rank == n else None)},
        # table-count independent work rate: (alive rows x 64-column panels) eliminated per second, whole job
        "row_panels_per_s": world * n * ((n + 63) // 64) / 2 / (elapsed / args.steps),
        "roofline": roofline,
    }
    if plain is not None:
        out["plain_ms_per_step"] = plain["ms_per_step"]
        out["parity_gate"]["plain_same_answer"] = plain["same_answer"]
    if target is not None:
        out[f"target_{args.target_n}"] = target
    if c4 is not None:
        out["batch_c4"] = c4
    if world == 1 and not args.no_extra_legs:
        # the reference's own example timings (examples/mt.py:16,32,35; examples/xoshiro.py) and the host-resident headline
        # system: extra fields, never `value`
        torch.cuda.empty_cache()
        out["c3_mt19937"] = c3_mt19937_leg(local_rank, not args.no_cpu_baseline)
        out["c5_xoshiro"] = c5_xoshiro_leg(local_rank, not args.no_cpu_baseline)
        out["h2d_inclusive"] = h2d_leg(n, seed, local_rank, dev)
    if world == 1 and not args.no_cpu_baseline:
        m = args.cpu_n or 65536
        cb = cpu_baseline(m, seed)
        out["parity_gate"]["gpu_equals_cpu_oracle_on_sample"] = gpu_equals_oracle(cb, m, seed, local_rank)
        out["cpu_baseline"] = cb
    return out


def target_leg(n: int, device: int, dev, ceil: dict) -> dict:
    """The north-star size (BASELINE.json: >= 70 % of the HBM roofline at 262144^2): 1 warm-up + 2 timed solve_one steps
    of the resident system, the same code path as the headline steps; extra field, never `value`."""
    stride = hip.padded_stride(n)
    free_b, _ = torch.cuda.mem_get_info(dev)
    need = 2.6 * n * stride * 8
    if free_b < need:
        return {"skipped": f"needs {need / 2**30:.1f} GiB of free HBM, {free_b / 2**30:.1f} available"}
    mat = torch.empty(n * stride, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    seed = FULL_RANK_SEEDS.get(n, 1234)               # full rank: the planted solution is the only one (bit-exactness by uniqueness)
    hip.synth_device(mat.data_ptr(), n, n, stride, seed, device=device, stream=stream)
    torch.cuda.synchronize(dev)
    steps = 2
    elapsed, stats, _ = timed_single(mat, n, stride, steps, 1, device, dev, 1, True)
    # the same steps again without the HIP-event brackets around the 2240 bulk launches (an event record is a barrier
    # packet: the next launch cannot ramp up under the tail of the previous one) -- what a caller of solve_one sees
    plain_elapsed, plain_stats, _ = timed_single(mat, n, stride, steps, 0, device, dev, 1, False)
    bad = hip.residual_device(mat.data_ptr(), n, n, stride, stats[-1].origin, device=device, stream=stream)
    del mat
    s0 = stats[-1].stats
    roofline, launches = roofline_block([s.stats for s in stats], float(sum(s.stats["ms_sweep"] for s in stats)), n, device, ceil)
    finish_roofline(roofline, s0, launches)
    elim_ms = float(sum(s.stats["ms_eliminate"] for s in stats))
    roofline["elimination_GBs"] = roofline["alg_bytes_total"] / (elim_ms * 1e-3) / 1e9
    roofline["elimination_frac"] = roofline["elimination_GBs"] / HBM_PEAK_GBS
    # what the bulk launches really moved through HBM (rows read once and written once per launch: K blocks per trip in an outer pass)
    # over the same time -- the figure in BYTES, beside the one in the roofline's unit (sweep-words, one per word and block applied)
    roofline["hbm_real_frac"] = roofline["hbm_bytes_total"] / (elim_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
    return {"n": n, "seed": seed, "steps": steps, "warmup": 1, "ms_per_step": elapsed / steps * 1e3,
            "ms_per_step_without_event_brackets": plain_elapsed / steps * 1e3,
            "same_answer_without_brackets": bool(all(np.array_equal(a.origin, b.origin) and a.rank == b.rank
                                                     for a, b in zip(stats, plain_stats))),
            "row_xors_per_s": float(sum(s.stats["row_xors"] for s in stats)) / elapsed,
            "rank": int(stats[-1].rank), "residual_rows": int(bad), "all_solved": all(s.solved for s in stats),
            "full_rank": bool(stats[-1].rank == n),
            # rank = N: A x = b has ONE solution, so equality with the generator's planted vector is bit-exactness (SURVEY hard part 7)
            "x_equals_planted": bool(all(np.array_equal(s.origin, hip.planted_solution(n, seed)) for s in stats + plain_stats)),
            "solve_wall_ms": {"eliminate": float(np.mean([s.stats["ms_eliminate"] for s in stats])),
                              "backsub": float(np.mean([s.stats["ms_backsub"] for s in stats]))},
            "roofline": roofline}


MT_VARIANTS = ((32, None), (17, None), (9, None), (1, None), (1337, 19968 // 1337 + 10), (137, 19968 // 137 + 60))


def c3_mt19937_leg(device: int, with_oracle: bool) -> dict:
    """BASELINE configs[2]: MT19937 state recovery, the six (bits per output, samples) variants the reference's examples/mt.py
    times (`generate system` :16,32 and `solve_one` :35), end to end through the list-of-int boundary the reference's own
    `m4ri_solve` takes: generate (symbolic MT19937 on BitVecs) . get_eqs . m4ri_solve (digits -> device pack kernel, elimination,
    back-substitution, export) . convert_sol, and the known answer of examples/mt.py:38 (the state of random.Random(3142)).
    Beside each: the CPU oracle on the same equation list (test infrastructure, here as the timed baseline only).  Warm = the
    second call of the same shape.  Never `value`."""
    import random

    from gf2bv_amd import LinearSystem, _internal
    from tests.harness_models import MT19937
    out, threads = [], None
    for bs, samples in MT_VARIANTS:
        rand = random.Random(3142)
        state = tuple(rand.getstate()[1][:-1])
        eff = ((bs - 1) & bs) or bs
        ns = 624 * 32 // eff if samples is None else samples
        obs = [rand.getrandbits(bs) for _ in range(ns)]
        lin = LinearSystem([32] * 624)
        mt = lin.gens()
        sym = MT19937(mt)
        t0 = time.perf_counter()
        zeros = [sym.getrandbits(bs) ^ o for o in obs] + [mt[0] ^ 0x80000000]
        t1 = time.perf_counter()
        eqs = lin.get_eqs(zeros)
        eqs += [0] * max(0, lin._cols - len(eqs))
        t2 = time.perf_counter()
        walls = []
        for _ in range(3):                                # cold (first call of this shape), then warm (the better of two)
            ta = time.perf_counter()
            raw = _internal.m4ri_solve(eqs, lin._cols, 0, device)
            walls.append(time.perf_counter() - ta)
        walls = [walls[0], min(walls[1:])]
        ok = raw is not None and lin.convert_sol(raw) == state
        # device-side split of the same call (the ctypes twin of the boundary: digits -> gf2bv_solve_digits)
        st = _digits_stats(eqs, lin._cols, device)
        rec = {"bits_per_output": bs, "outputs": ns, "rows": len(eqs), "cols": lin._cols,
               "generate_ms": (t1 - t0) * 1e3, "get_eqs_ms": (t2 - t1) * 1e3,
               "m4ri_solve_ms": {"first": walls[0] * 1e3, "warm": walls[1] * 1e3},
               "device_ms": st, "recovered_state_equals_known_answer": bool(ok)}
        if with_oracle:
            cb = cpu_baseline_list(eqs, lin._cols, 0, 1)
            rec["cpu_oracle_ms"] = cb["ms_median"]
            rec["gpu_equals_cpu_oracle"] = bool(cb["result"] == raw)
            threads = cb["threads"]
        out.append(rec)
    # a batch of such recoveries -- one system per instance -- through m4ri_solve_many (lock-step gangs; the digits in chunks of whole
    # systems, gathered under the previous chunk's solve, round 5) beside the loop of single calls
    nb = 8
    systems, states = [], []
    for k in range(nb):
        rand = random.Random(3142 + k)
        states.append(tuple(rand.getstate()[1][:-1]))
        obs = [rand.getrandbits(32) for _ in range(624)]
        lin = LinearSystem([32] * 624)
        mt = lin.gens()
        sym = MT19937(mt)
        eqs = lin.get_eqs([sym.getrandbits(32) ^ o for o in obs] + [mt[0] ^ 0x80000000])
        systems.append(eqs + [0] * max(0, lin._cols - len(eqs)))
    walls = []
    for _ in range(3):
        ta = time.perf_counter()
        many = _internal.m4ri_solve_many(systems, lin._cols, 0, device)
        walls.append(time.perf_counter() - ta)
    ta = time.perf_counter()
    loop = [_internal.m4ri_solve(e, lin._cols, 0, device) for e in systems]
    tl = time.perf_counter() - ta
    batch = {"systems": nb, "bits_per_output": 32, "rows": len(systems[0]), "cols": lin._cols,
             "m4ri_solve_many_ms_per_system": {"first": walls[0] * 1e3 / nb, "warm": min(walls[1:]) * 1e3 / nb},
             "loop_of_m4ri_solve_ms_per_system": tl * 1e3 / nb,
             "recovered_states_equal_known_answers": bool(many == loop and all(r is not None and lin.convert_sol(r) == st
                                                                              for r, st in zip(many, states)))}
    return {"workload": "MT19937 state recovery (19968 unknowns), the six variants of the reference's examples/mt.py, list-of-int boundary",
            "cpu_oracle_threads": threads if with_oracle else None, "variants": out, "batch": batch}


def _digits_stats(eqs, cols: int, device: int) -> dict:
    """pack / H2D . elimination . back-substitution . export of one solve through gf2bv_solve_digits (what m4ri_solve calls)"""
    nd = (cols + 1 + 31) // 32                             # 32-bit digits, the same count for every equation
    mask = (1 << (32 * nd)) - 1
    dig = np.frombuffer(b"".join((abs(e) & mask).to_bytes(4 * nd, "little") for e in eqs), dtype=np.uint32)
    off = np.arange(len(eqs) + 1, dtype=np.int64) * nd
    s = hip.solve_digits(dig, off, 32, len(eqs), cols, hip.MODE_SINGLE, device=device).stats
    # fast_blocks: blocks of 256 pivots factorised by a one-launch search (round 5: k_block_sparse on these systems; the general panel
    # steps took every block until round 4: 5 launches of 21-55 us each)
    return {"pack_h2d": s["ms_pack"], "eliminate": s["ms_eliminate"], "backsub": s["ms_backsub"], "export": s["ms_export"],
            "total_host": s["ms_total"], "fast_blocks": int(s["fast_blocks"]), "blocks": (cols + 255) // 256}


def c5_xoshiro_leg(device: int, with_oracle: bool) -> dict:
    """BASELINE configs[4]: xoshiro256** seed recovery with solve_all (640 x 256, unique solution, kernel-basis path), the
    scenario of the reference's examples/xoshiro.py: cold = the first solve of this shape in the process, warm = median of 200."""
    import random

    from gf2bv_amd import LinearSystem
    from tests.harness_models import Xoshiro256starstar
    rnd = random.Random(1)
    gen = Xoshiro256starstar([rnd.getrandbits(64) for _ in range(4)])
    secret = tuple(gen.s)
    outs = [gen() for _ in range(10)]
    lin = LinearSystem([64] * 4)
    sym = Xoshiro256starstar(lin.gens())
    t0 = time.perf_counter()
    zeros = [sym.step() ^ Xoshiro256starstar.untemper(o) for o in outs]
    t1 = time.perf_counter()
    sols = list(lin.solve_all(zeros))
    cold = time.perf_counter() - t1
    warm = []
    for _ in range(200):
        ta = time.perf_counter()
        again = list(lin.solve_all(zeros))
        warm.append(time.perf_counter() - ta)
    warm.sort()
    rec = {"workload": "xoshiro256** seed recovery, solve_all, 640 x 256 (reference examples/xoshiro.py)",
           "generate_ms": (t1 - t0) * 1e3, "solve_all_ms": {"cold": cold * 1e3, "warm_median": warm[len(warm) // 2] * 1e3,
                                                            "warm_p99": warm[int(len(warm) * .99)] * 1e3},
           "solutions": len(sols), "recovered_seed_equals_known_answer": bool(sols == [secret] and again == sols)}
    if with_oracle:
        eqs = lin.get_eqs(zeros)
        eqs += [0] * max(0, 256 - len(eqs))
        cb = cpu_baseline_list(eqs, 256, 1, 21)
        rec["cpu_oracle_ms_median"] = cb["ms_median"]
        rec["gpu_equals_cpu_oracle"] = bool([lin.convert_sol(s) for s in cb["result"]] == sols)
    return rec


def h2d_leg(n: int, seed: int, device: int, dev) -> dict:
    """The headline system starting on the HOST (SURVEY 8d phase list: pack/H2D . elimination . back-subst . export): the same
    65536^2 solve through gf2bv_solve_words, i.e. from a row-major uint64 buffer in host memory (pageable, as a caller's numpy
    array is).  Never `value` (inputs of the timed region of `value` are resident in HBM)."""
    stride = hip.padded_stride(n)
    buf = hip.DeviceBuffer(n * stride * 8, device)
    hip.synth_device(buf.ptr, n, n, stride, seed, device=device)
    host = buf.download().view(np.uint64).reshape(n, stride)
    buf.free()
    walls, st = [], None
    for _ in range(2):
        ta = time.perf_counter()
        sol = hip.solve_words(host, n, n, hip.MODE_SINGLE, device=device)
        walls.append(time.perf_counter() - ta)
        st = sol.stats
    ok = bool(sol.rank == n and np.array_equal(sol.origin, hip.planted_solution(n, seed))) if n in FULL_RANK_SEEDS else None
    return {"n": n, "input": f"{n * stride * 8 / 2**20:.0f} MiB row-major uint64 in pageable host memory",
            "ms_per_solve": {"first": walls[0] * 1e3, "second": walls[1] * 1e3},
            "phases_ms": {"pack_h2d": st["ms_pack"], "eliminate": st["ms_eliminate"], "backsub": st["ms_backsub"],
                          "export": st["ms_export"], "total_host": st["ms_total"]},
            "h2d_GBs": n * stride * 8 / (st["ms_pack"] * 1e-3) / 1e9 if st["ms_pack"] > 0 else None,
            "x_equals_planted": ok}


def batch_job(args, world, rank, local_rank, dev, steps: int, warmup: int):
    """BASELINE configs[3]: args.batch_total independent n x n systems, contiguous blocks per rank, one gather.
    `warmup` untimed + `steps` timed passes of the whole job; returns the measurement dict on rank 0, None elsewhere."""
    n, total = args.batch_n, args.batch_total
    lo, hi = batch.shard_bounds(total, world, rank)
    seeds = [BATCH_SEED0 + i for i in range(lo, hi)]
    stride = hip.padded_stride(n)
    mats = batch.synth_shard(n, seeds, local_rank)             # resident in HBM before the timed region
    torch.cuda.synchronize(dev)
    kernel_events = not args.no_kernel_events

    def step():
        recs, sols = batch.solve_shard(n, mats, local_rank, time_kernels=kernel_events)
        allrec = batch.gather_records(recs, total)              # the single end-of-job collective (no-op at N = 1)
        return allrec, sols

    for _ in range(warmup):
        step()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    allrec, sols, all_sols = None, None, []
    for _ in range(steps):
        allrec, sols = step()
        all_sols.extend(sols)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # parity gate: every local system's A x = b on the untouched input (independent kernel); the gathered table must
    # hold this rank's records at its block and report every system of the job solved
    bad = sum(hip.residual_device(mats[i].data_ptr(), n, n, stride, s.origin, device=local_rank)
              for i, s in enumerate(sols))
    rec_ok = bool(allrec.shape[0] == total and int((allrec[:, 0] != 0).sum().item()) == 0)
    mine = np.stack([batch.make_record(s.status, s.rank, s.origin) for s in sols]) if sols else None
    if mine is not None:
        rec_ok = rec_ok and bool(np.array_equal(allrec[lo:hi].cpu().numpy(), mine))
    ok = torch.tensor([1 if (bad == 0 and rec_ok and all(s.solved for s in sols)) else 0], device=dev)
    agg = torch.tensor([float(sum(s.stats["row_xors"] for s in all_sols)),
                        float(sum(16.0 * s.stats["sweep_words"] for s in all_sols)),
                        # a gang's launches serve all its systems and every member reports the gang's time
                        float(sum(s.stats["ms_sweep"] / max(s.stats.get("gang_systems", 1), 1) for s in all_sols)),
                        float(sum(s.stats["n_sweeps"] / max(s.stats.get("gang_systems", 1), 1) for s in all_sols))],
                       dtype=torch.float64, device=dev)
    per_rank = torch.tensor([float(hi - lo) * steps / elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(agg)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    del mats
    torch.cuda.empty_cache()
    if rank != 0:
        return None
    row_xors, alg_bytes, sweep_ms, launches = (float(x) for x in agg.tolist())
    s0 = sols[0].stats
    ceil = hip.stream_ceiling(2 << 30, local_rank)
    roofline = None
    if sweep_ms > 0:
        achieved = alg_bytes / (sweep_ms * 1e-3) / 1e9
        g = s0["panels_per_sweep"]
        roofline = {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            # PMC (profiles/r04_batch_pmc.txt): HBM bytes of an average launch of one gang, scaled to this gang size
            "traffic": pmc_traffic(n, g, s0["tables_per_sweep"] // g, "k_update16", int(s0.get("gang_systems", 0))),
            "kernel": f"k_update16<G={g},T={s0['tables_per_sweep'] // g}> (bulk update of a gang: "
                      f"{s0.get('gang_systems', 0)} systems x {64 * g} pivots per launch)",
            "launches": launches, "alg_bytes_per_launch": alg_bytes / max(launches, 1),
            "avg_launch_ms": sweep_ms / max(launches, 1),
            "persistent_rmw_stream_GBs": ceil["rmw_gbs"], "persistent_read_stream_GBs": ceil["read_gbs"],
            # end to end: the job's algorithmic bytes over its wall time (all ranks), against N x 8 TB/s
            "end_to_end_frac": alg_bytes / steps / (elapsed / steps) / 1e9 / (HBM_PEAK_GBS * world),
            "note": "summed over all ranks; two gangs are in flight per GPU (a gang's back-substitution and export overlap "
                    "the next gang's elimination), so a launch's duration includes the share of the chip the other gang took",
        }
    return {
        "value": row_xors / elapsed, "unit": "row-XORs/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": elapsed / steps * 1e3, "scaling": "strong",
        "config": {
            "workload": f"batch of {total} independent synthetic dense {n}x{n} GF(2) solve_one systems (seeds {BATCH_SEED0}..), "
                        f"contiguous blocks per GPU, one all_gather of [status, rank, origin] records at the end",
            "systems_total": total, "systems_per_gpu": (total + world - 1) // world, "n": n,
            "parallelism": f"independent systems, shard x{world}, gangs of {s0.get('gang_systems', 0)}",
            "tables_per_sweep": s0["tables_per_sweep"], "table_bits": s0["table_bits"], "tile_words": s0["tile_words"],
            "collective": ("all_gather over nccl (RCCL)" if world > 1 else "none (single rank)"),
        },
        "systems_per_s": total * steps / elapsed,
        "ms_per_system_per_gpu": elapsed / steps / ((total + world - 1) // world) * 1e3,
        "rank0_systems_per_s": float(per_rank.item()),
        "parity_gate": {"residual_rows_rank0": int(bad), "all_ranks_ok": bool(ok.item()),
                        "gathered_records": int(allrec.shape[0])},
        "row_panels_per_s": total * steps * n * ((n + 63) // 64) / 2 / elapsed,
        "roofline": roofline,
    }


def run_batch(args, world, rank, local_rank, dev):
    """`--workload batch`: the configs[3] job as the line's own value."""
    job = batch_job(args, world, rank, local_rank, dev, args.steps, args.warmup)
    if rank != 0:
        return None
    out = {"metric": "GF(2) row-XORs/s (batch of independent dense NxN solve_one, sharded)"}
    out.update(job)
    out.update({"higher_is_better": True, "vs_baseline": None, "dtype": "u64", "data": "synthetic"})
    if world == 1 and not args.no_cpu_baseline:
        m = args.cpu_n or args.batch_n
        cb = cpu_baseline(m, BATCH_SEED0)
        out["parity_gate"]["gpu_equals_cpu_oracle_on_sample"] = gpu_equals_oracle(cb, m, BATCH_SEED0, local_rank)
        out["cpu_baseline"] = cb
    return out


def run_sharded(args, world, rank, local_rank, dev):
    """ONE dense N x N system, column tiles cyclic over the ranks (gf2bv_amd.slab): per block one broadcast of the
    block's records from the owner of its window, bulk update of the own tiles on every rank, tiles collected on rank 0."""
    from gf2bv_amd import slab
    n = args.n
    seed = args.seed if args.seed is not None else FULL_RANK_SEEDS.get(n, 1234)
    stride = hip.padded_stride(n)
    mat = torch.empty(n * stride, dtype=torch.int64, device=dev)       # the same system on every rank
    stream = torch.cuda.current_stream(dev).cuda_stream
    hip.synth_device(mat.data_ptr(), n, n, stride, seed, device=local_rank, stream=stream)
    torch.cuda.synchronize(dev)
    if world == 1 and not dist.is_initialized():                       # the schedule talks to a process group
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29544")
        dist.init_process_group("gloo", rank=0, world_size=1)
    sols = []
    for _ in range(args.warmup):
        slab.solve_one_sharded(mat, n, n, stride, local_rank)
    torch.cuda.synchronize(dev)
    dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sols.append(slab.solve_one_sharded(mat, n, n, stride, local_rank))
    torch.cuda.synchronize(dev)
    dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank != 0:
        return None
    bad = hip.residual_device(mat.data_ptr(), n, n, stride, sols[-1].origin, device=local_rank, stream=stream)
    s0 = sols[-1].stats
    return {
        "metric": "GF(2) row-XORs/s (ONE dense NxN solve_one, columns sharded over the GPUs)",
        "value": float(sum(s.stats["row_xors"] for s in sols)) / elapsed, "unit": "row-XORs/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": f"ONE synthetic dense {n}x{n} GF(2) solve_one, seed {seed}, column tiles cyclic over {world} GPUs, "
                               f"one broadcast of {n * 32 / 2**20:.1f} MiB of multipliers + block records per block of 256 pivots, "
                               f"tiles collected on rank 0 for the back-substitution",
                   "parallelism": f"column slabs x{world}", "tables_per_sweep": s0["tables_per_sweep"],
                   "tile_words": s0["tile_words"], "rank": int(sols[-1].rank)},
        "parity_gate": {"residual_rows": int(bad), "all_solved": all(s.solved for s in sols)},
        "roofline": None,
    }


def dry_run_ranks(args, world: int, rank: int) -> None:
    """Every rank computes its own plan of the configs[3] job, the plans are gathered (gloo: no GPU is touched) and rank 0 prints
    them with the checks a real run relies on: the blocks tile [0, total) in rank order, no two ranks write the same rows of the
    gathered table, and every rank's resident bytes fit its GPU."""
    if world > 1:
        dist.init_process_group("gloo")
    plan = batch.rank_plan(args.batch_total, args.batch_n, world, rank)
    plans = [plan]
    if world > 1:
        plans = [None] * world
        dist.all_gather_object(plans, plan)
    if rank == 0:
        cover = sorted(tuple(p["systems"]) for p in plans)
        ok_cover = cover[0][0] == 0 and cover[-1][1] == args.batch_total and all(a[1] == b[0] for a, b in zip(cover, cover[1:]))
        ok_gangs = all(p["gangs"] == [] or (p["gangs"][0][0] == p["systems"][0] and p["gangs"][-1][1] == p["systems"][1]
                                             and all(a[1] == b[0] for a, b in zip(p["gangs"], p["gangs"][1:]))) for p in plans)
        ok_mem = all(p["resident_bytes"]["peak_estimate"] + p["resident_bytes"]["pool_cap"] <= p["hbm_bytes"] for p in plans)
        print(json.dumps({"dry_run_ranks": True, "workload": f"{args.batch_total} x {args.batch_n}^2 independent systems over {world} ranks",
                          "collective": "ONE all_gather of [status, rank, origin] records at the end (RCCL in a real run)",
                          "blocks_tile_the_job": bool(ok_cover), "gangs_tile_each_block": bool(ok_gangs), "fits_in_hbm": bool(ok_mem),
                          "ranks": plans}), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.dry_run_ranks:
        return dry_run_ranks(args, world, rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (the product path has no CPU fallback)")
    # (tests on a one-GPU box: GF2BV_BENCH_DEVICE pins every rank to one GPU, GF2BV_BENCH_BACKEND=gloo replaces RCCL, which
    # cannot put two ranks on one device; the driver's multi-GPU runs set neither)
    if os.environ.get("GF2BV_BENCH_DEVICE") is not None:
        local_rank = int(os.environ["GF2BV_BENCH_DEVICE"])
    backend = os.environ.get("GF2BV_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    rank_info = None
    if world > 1:
        # watchdog + one diagnostic line per rank (device, PCI id, RCCL version, seconds the init took) on stderr BEFORE anything is
        # allocated; an init that does not return within the limit ends the rank with exit code 3 and the RCCL log -- never a hang
        rank_info = batch.init_process_group_guarded(backend, dev if backend == "nccl" else None,
                                                     limit_s=float(os.environ.get("GF2BV_BENCH_INIT_LIMIT_S", "240")))
    print(f"[gf2bv bench] rank {rank}: libgf2bv_hip.so build {hip.build_id()}", file=sys.stderr, flush=True)
    workload = args.workload if args.workload != "auto" else "single"
    out = {"single": run_single, "batch": run_batch, "sharded": run_sharded}[workload](args, world, rank, local_rank, dev)
    if rank == 0:
        out["build_id"] = hip.build_id()
        if rank_info is not None:
            out["rank0"] = rank_info
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
