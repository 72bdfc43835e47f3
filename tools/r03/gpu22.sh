#!/bin/bash
# page-phased k_update16k (tables of one page rebuilt behind the other page's lookups): isolated rate, parity, in-solve times
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd $R
{ for K in 2 4 8; do ./tools/_mbk_bin 131072 1024 $K; done; ./tools/_mbk_bin 262144 512 8; } > $O/r03_mbk22.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_stress.py -x -q -k "two_level or outer" > $O/r03_pytest22a.log 2>&1; echo "stress two-level rc=$?" > $O/r03_pytest22.summary
{ for n in 131072 196608 262144; do python tools/profile_one.py $n 3 | tail -2; done; } > $O/r03_times22.txt 2>&1
GF2BV_TWO_LEVEL=8 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q > $O/r03_pytest22b.log 2>&1; echo "parity K=8 rc=$?" >> $O/r03_pytest22.summary
