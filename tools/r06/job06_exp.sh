#!/bin/bash
# EXPERIMENT: upper bound of what a look-ahead fused into the block launch could save (gate + k_prio_window skipped: WRONG answers, timing only)
run() { echo "## $*"; env "$@" RESIDUAL=0 python tools/profile_one.py 65536 4 | tail -2 | cut -c1-120; env "$@" RESIDUAL=0 python tools/profile_one.py 32768 4 | tail -2 | cut -c1-120; }
( run GF2BV_X=0; run GF2BV_EXP_SKIP_PRIO=0; run GF2BV_EXP_SKIP_PRIO=64; run GF2BV_X=0 ) > gpurun_out/r06f_exp.txt 2>&1
cat gpurun_out/r06f_exp.txt
python tools/r06/rccl_init_ab.py "lease $(date +%H%M)" >> gpurun_out/r06_rccl_init.txt 2>&1
tail -8 gpurun_out/r06_rccl_init.txt
