#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
{
for t in 1 4; do echo "threads $t"; GF2BV_BATCH_THREADS=$t timeout 300 python tools/mt_batch_digits_time.py 16 32; done
echo gangs; GF2BV_SPARSE_BATCH=0 timeout 300 python tools/mt_batch_digits_time.py 16 32
} > gpurun_out/r05_mt_digits.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl" | tail -n 5 >> gpurun_out/r05_mt_digits.txt
