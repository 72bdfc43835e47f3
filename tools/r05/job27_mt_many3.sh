#!/bin/bash
# m4ri_solve_many in chunks with recycled page-locked staging: MT19937 batches again, and the parity files
cd /root/repo; mkdir -p gpurun_out
{
timeout 300 python tools/mt_many_time.py 8 32
timeout 300 python tools/mt_many_time.py 16 32
timeout 300 python tools/mt_many_time.py 32 32
GF2BV_BATCH_CHUNK_MB=1024 timeout 300 python tools/mt_many_time.py 32 32
GF2BV_BATCH_CHUNK_MB=4096 GF2BV_HOST_POOL_MB=9000 timeout 300 python tools/mt_many_time.py 32 32
for bs in 17 9 1; do timeout 600 python tools/mt_many_time.py 8 $bs | tail -n 1; done
timeout 300 python tools/mt_batch_digits_time.py 16 32
} > gpurun_out/r05_mt_many3.txt 2>&1
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch_c4.py -m gpu -x -q 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl" | tail -n 5 >> gpurun_out/r05_mt_many3.txt
