#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd $R
KEEP_TRACE=1 bash tools/jobs/kernel_stats.sh r04_65536t python tools/profile_one.py 65536 3
