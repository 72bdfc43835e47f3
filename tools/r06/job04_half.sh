#!/bin/bash
# Round 6: the outer pass with half-block tables, two workgroups per CU (GF2BV_OUTER_HALF=1) -- parity under forced two-level plans, then A/B.
mkdir -p gpurun_out
( GF2BV_OUTER_HALF=1 timeout 900 python -m pytest tests/test_gpu_stress.py tests/test_gpu_parity.py -x -q -m gpu -k "two_level or outer or target_262144 or pivotless" 2>&1 | tail -5 ) > gpurun_out/r06d_pytest.log 2>&1
cat gpurun_out/r06d_pytest.log
run() { echo "## $*"; env "$@" SEED=1242 python tools/profile_one.py 262144 3 | tail -2 | cut -c1-150; env "$@" python tools/profile_one.py 131072 3 | tail -2 | cut -c1-150; }
( run GF2BV_OUTER_HALF=1; run GF2BV_OUTER_HALF=0; run GF2BV_OUTER_HALF=1; run GF2BV_OUTER_HALF=0 ) > gpurun_out/r06d_half_ab.txt 2>&1
cat gpurun_out/r06d_half_ab.txt
