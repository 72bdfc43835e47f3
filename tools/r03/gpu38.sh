#!/bin/bash
# slabs padded to whole KiB (64 rows): whole suite, stress, bench
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -m gpu -x -q > $O/r03_pytest38.log 2>&1; echo "full suite rc=$?" > $O/r03_pytest38.summary
GF2BV_TWO_LEVEL=8 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q > $O/r03_pytest38b.log 2>&1; echo "parity K=8 rc=$?" >> $O/r03_pytest38.summary
timeout 500 python tests/manual/stress_parity.py 200 47 > $O/r03_stress38.log 2>&1; echo "stress rc=$?" >> $O/r03_pytest38.summary
python bench.py > $O/r03_bench_default.json 2> $O/r03_bench_default.err
