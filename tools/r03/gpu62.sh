#!/bin/bash
# outer panels of more than 8 blocks (GF2_KMAX = 12)
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd $R
{ for K in 8 10 12; do ./tools/_mbk 131072 1024 $K; done; } > $O/r03_k12_62.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_stress.py -x -q -k "two_level" > $O/r03_pytest62.log 2>&1; echo "two-level stress rc=$?" > $O/r03_final62.summary
{ for K in 8 12 10 8 12; do echo "## GF2BV_OUTER_K=$K"; for n in 131072 196608 262144; do GF2BV_OUTER_K=$K timeout 200 python tools/profile_one.py $n 3 | tail -1; done; done; } >> $O/r03_k12_62.txt 2>&1
