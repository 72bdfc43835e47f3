#!/bin/bash
# Late round 5: the A/B loops behind profiles/r05_outer_shapes.txt (sections 8-15), on the knobs the final build keeps.
# usage (GPU box, repo root): bash tools/r05/job43_outer_pass_ab.sh > gpurun_out/outer_pass_ab.txt
# (Sections 2-7 compared kernels that no longer exist as instances -- <8,1024,4>, <10,768,4>, budgets 104 / 112 / 128, 10-14 segments:
#  rebuild them from update16k_body with GF2_U16K_WIDE(name, SEG, RB, NB, budget / 2) and a case in enqueue_outer_apply.)
run() { echo "## $*"; env "$@" SEED=1242 python tools/profile_one.py 262144 3 | tail -2 | cut -c1-110; env "$@" python tools/profile_one.py 131072 3 | tail -2 | cut -c1-110; }
run GF2BV_OUTER_SHAPE=0                                    # the default: 16 wavefronts x 12 segments @ 120 registers, chunk-major, side launch
run GF2BV_OUTER_SHAPE=1 GF2BV_OUTER_ORDER=0 GF2BV_OUTER_SIDE=0   # rounds 3-5: 8 wavefronts x 16 segments, tile-major, everything on the outer stream
run GF2BV_OUTER_SHAPE=1                                    # the old shape under the new order / side launch
run GF2BV_OUTER_SHAPE=2                                    # 16 x 10 @ 120 (no scratch)
run GF2BV_OUTER_SHAPE=3                                    # 16 x 12 @ 112 (k_block_fast_narrow fits beside it)
run GF2BV_OUTER_ORDER=0                                    # tile-major items
run GF2BV_OUTER_XCD=1                                      # round 4's chunk-major order per XCD
run GF2BV_OUTER_SIDE=0                                     # the next panel's tiles in front of the pass
run GF2BV_OUTER_SPLIT=50                                   # the pass over two streams (opt-in)
for K in 8 10 12; do run GF2BV_OUTER_K=$K; done
for mib in 256 384 512 768; do run GF2BV_TWO_LEVEL_MIN_MIB=$mib; done
for o in 0 2; do MB_ORDER=$o tools/_probe/mbk_wide 262144 1024 12; MB_ORDER=$o tools/_probe/mbk_legacy 262144 1024 12; done 2>/dev/null   # (hipcc -DMB_WIDE / default of tools/microbench_update16k.hip)
