#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_stress.py -x -q -k "sparse_block" > $O/r05_job6_tests.log 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "mt19937 or golden or sparse or density" >> $O/r05_job6_tests.log 2>&1
python tools/mt_stats.py 32 17 9 1 1337 137 > $O/r05_mt_stats_sparse.txt 2>&1
GF2BV_SPARSE_FAST=0 python tools/mt_stats.py 32 1 >> $O/r05_mt_stats_sparse.txt 2>&1
JOB_TIMEOUT=600 bash tools/jobs/kernel_stats.sh r05_c3_mt32_sparse python tools/mt_stats.py 32
JOB_TIMEOUT=600 bash tools/jobs/kernel_stats.sh r05_c3_mt1_sparse python tools/mt_stats.py 1
