#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd $R
{ for tl in 0 "" 8; do echo "## GF2BV_TWO_LEVEL=$tl"; GF2BV_TWO_LEVEL=$tl timeout 120 python tools/profile_one.py 131072 3 | tail -2; GF2BV_TWO_LEVEL=$tl timeout 200 python tools/profile_one.py 262144 3 | tail -2; done; } > $O/r03_two_level_times6.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q > $O/r03_pytest14.log 2>&1; echo "full suite rc=$?" > $O/r03_pytest14.summary
GF2BV_TWO_LEVEL=8 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stress.py -x -q > $O/r03_pytest14b.log 2>&1; echo "K=8 rc=$?" >> $O/r03_pytest14.summary
python bench.py > $O/r03_bench_default.json 2> $O/r03_bench_default.err
