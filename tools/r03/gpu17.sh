#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd $R
{ for tl in "" 8 0; do echo "## GF2BV_TWO_LEVEL=$tl"; GF2BV_TWO_LEVEL=$tl timeout 120 python tools/profile_one.py 131072 3 | tail -2; GF2BV_TWO_LEVEL=$tl timeout 200 python tools/profile_one.py 262144 3 | tail -2; done
  echo "## default at 65536 / 98304 / 114688 / 196608"; for n in 65536 98304 114688 196608; do python tools/profile_one.py $n 3 | tail -1; GF2BV_TWO_LEVEL=0 python tools/profile_one.py $n 3 | tail -1; done; } > $O/r03_two_level_times9.txt 2>&1
timeout 1800 python -m pytest tests -m gpu -x -q > $O/r03_pytest17.log 2>&1; echo "full suite rc=$?" > $O/r03_pytest17.summary
GF2BV_TWO_LEVEL=8 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q > $O/r03_pytest17b.log 2>&1; echo "parity K=8 rc=$?" >> $O/r03_pytest17.summary
timeout 900 python tests/manual/stress_parity.py 240 31 > $O/r03_stress17.log 2>&1; echo "stress rc=$?" >> $O/r03_pytest17.summary
