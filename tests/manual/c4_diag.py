"""Diagnostic: one GPU's share of the batch workload, printing rank / residual / hand-overs per system."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gf2bv_amd import batch, hip
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
nsys = int(sys.argv[2]) if len(sys.argv) > 2 else 64
tk = (sys.argv[3] == "1") if len(sys.argv) > 3 else True
seeds = [5000 + 192 + i for i in range(nsys)]
mats = batch.synth_shard(n, seeds, 0)
recs, sols = batch.solve_shard(n, mats, 0, time_kernels=tk)
stride = hip.padded_stride(n)
bad = 0
for i, s in enumerate(sols):
    r = hip.residual_device(mats[i].data_ptr(), n, n, stride, s.origin)
    if r or s.rank > n:
        bad += 1
        print("system", i, "rank", s.rank, "residual", r, "handovers", s.stats["search_handovers"])
print("bad systems:", bad, "of", nsys, " handovers total", sum(s.stats["search_handovers"] for s in sols),
      " env SELF_WAIT", os.environ.get("GF2BV_SELF_WAIT_US"))
