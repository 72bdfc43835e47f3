#!/bin/bash
# round 4: the outer pass chunk-major per XCD (GF2BV_OUTER_XCD), thresholds, streaming row accesses for single systems; parity of the two-level paths
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_stress.py tests/test_gpu_parity.py -x -q -k "two_level or 262144 or large_dense or beyond or gang" > $O/r04_pytest08.log 2>&1; echo "rc=$?" > $O/r04_gpu08.summary
{
for cfg in "GF2BV_OUTER_XCD=1" "GF2BV_OUTER_XCD=0" "GF2BV_OUTER_XCD=1" "GF2BV_OUTER_XCD=0" ; do
  echo "## $cfg"; for n in 131072 262144; do env $cfg timeout 300 python tools/profile_one.py $n 3 | tail -2; done
done
for cfg in "GF2BV_TWO_LEVEL_MIN_MIB=256" "GF2BV_TWO_LEVEL_MIN_MIB=384" "GF2BV_TWO_LEVEL_MIN_MIB=768" "GF2BV_OUTER_K=8" "GF2BV_OUTER_K=10" "GF2BV_SINGLE_NT=1" "GF2BV_SINGLE_NT=1 GF2BV_TWO_LEVEL=0"; do
  echo "## $cfg"; for n in 262144; do env $cfg timeout 300 python tools/profile_one.py $n 3 | tail -2; done
done
for cfg in "" "GF2BV_SINGLE_NT=1"; do
  echo "## $cfg"; for n in 65536 131072; do env $cfg timeout 300 python tools/profile_one.py $n 4 | tail -2; done
done
} > $O/r04_target_ab08.txt 2>&1
