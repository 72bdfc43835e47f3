#!/bin/bash
# final default bench line of the round (current build)
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd $R
python bench.py > $O/r03_bench_default.json 2> $O/r03_bench_default.err
timeout 600 python -m pytest tests/test_gpu_batch_c4.py tests/test_gpu_stress.py -x -q > $O/r03_pytest27.log 2>&1; echo "bench + stress tests rc=$?" > $O/r03_pytest27.summary
