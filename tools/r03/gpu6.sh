#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd $R
for k in 2 4 3; do GF2BV_TWO_LEVEL=$k timeout 900 python -m pytest tests/test_gpu_parity.py -x -q > $O/r03_pytest6_k$k.log 2>&1; echo "K=$k rc=$?" >> $O/r03_pytest6.summary; done
{ for tl in 0 "" 2 4 8; do echo "## GF2BV_TWO_LEVEL=$tl"; for n in 65536 131072; do GF2BV_TWO_LEVEL=$tl python tools/profile_one.py $n 3 | tail -2; done; GF2BV_TWO_LEVEL=$tl python tools/profile_one.py 262144 2 | tail -1; done; } > $O/r03_two_level_times.txt 2>&1
