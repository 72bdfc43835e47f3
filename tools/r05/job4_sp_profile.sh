#!/bin/bash
O=gpurun_out; mkdir -p $O
KEEP_TRACE=1 SEED=1242 JOB_TIMEOUT=600 bash tools/jobs/kernel_stats.sh r05_sp262144 python tools/profile_one.py 262144 1
python tools/sp_timeline.py $O/r05_sp262144_trace > $O/r05_sp262144_timeline.txt 2>&1
KEEP_TRACE=1 SEED=1242 GF2BV_STRASSEN=0 JOB_TIMEOUT=600 bash tools/jobs/kernel_stats.sh r05_sp262144_L0 python tools/profile_one.py 262144 1
python tools/sp_timeline.py $O/r05_sp262144_L0_trace > $O/r05_sp262144_L0_timeline.txt 2>&1
find $O -name "*kernel_trace.csv" -delete
