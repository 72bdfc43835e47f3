#!/bin/bash
# m4ri_solve_many on MT19937 batches: system by system on host threads (new) against the lock-step gangs
cd /root/repo; mkdir -p gpurun_out
{
for t in 1 2 4 8; do GF2BV_BATCH_THREADS=$t timeout 300 python tools/mt_many_time.py 32 32 | tail -n 1; done
GF2BV_SPARSE_BATCH=0 timeout 300 python tools/mt_many_time.py 32 32 | tail -n 1
for t in 4 8; do GF2BV_BATCH_THREADS=$t timeout 300 python tools/mt_many_time.py 8 32 | tail -n 1; done
for bs in 17 9 1; do timeout 600 python tools/mt_many_time.py 8 $bs | tail -n 1; done
} > gpurun_out/r05_mt_many2.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_batch_c4.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -n 3 >> gpurun_out/r05_mt_many2.txt
