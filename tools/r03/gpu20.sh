#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd $R
{ echo "## time_kernels off / on at 262144 and 131072"; for tk in 0 1; do TIME_KERNELS=$tk python tools/profile_one.py 262144 3 | tail -2; TIME_KERNELS=$tk python tools/profile_one.py 131072 3 | tail -2; done; } > $O/r03_times20.txt 2>&1
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r03_stats_262144 -- python $R/tools/profile_one.py 262144 1 > $O/r03_stats_262144.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r03_stats_65536 -- python $R/tools/profile_one.py 65536 1 > $O/r03_stats_65536.log 2>&1
cd $R; timeout 1800 python -m pytest tests -m gpu -x -q > $O/r03_pytest20.log 2>&1; echo "full suite rc=$?" > $O/r03_pytest20.summary
