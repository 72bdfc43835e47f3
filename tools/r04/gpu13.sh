#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_small.py tests/test_gpu_stress.py -x -q > $O/r04_pytest13.log 2>&1; echo "rc=$?" > $O/r04_gpu13.summary
