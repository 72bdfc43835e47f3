#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -m gpu -x -q > $O/r03_pytest64.log 2>&1; echo "full suite rc=$?" > $O/r03_final64.summary
GF2BV_TWO_LEVEL=12 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q > $O/r03_pytest64b.log 2>&1; echo "parity K=12 rc=$?" >> $O/r03_final64.summary
GF2BV_TWO_LEVEL=5 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "not 262144 and not 327680" > $O/r03_pytest64c.log 2>&1; echo "parity K=5 rc=$?" >> $O/r03_final64.summary
timeout 400 python tests/manual/stress_parity.py 240 61 > $O/r03_stress64.log 2>&1; echo "stress rc=$?" >> $O/r03_final64.summary
{ for n in 65536 131072 196608 262144; do timeout 200 python tools/profile_one.py $n 3 | tail -1; done; timeout 300 python tools/largest_run.py | tail -1; } > $O/r03_times64.txt 2>&1
