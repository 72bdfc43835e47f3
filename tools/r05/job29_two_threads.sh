#!/bin/bash
# digits batches on two host threads with per-gang uploads: MT19937 batches, the parity / batch files, the gang stress
cd /root/repo; mkdir -p gpurun_out
{
timeout 300 python tools/mt_batch_digits_time.py 16 32
GF2BV_BATCH_THREADS=1 timeout 300 python tools/mt_batch_digits_time.py 16 32 | tail -n 3
GF2BV_BATCH_THREADS=3 timeout 300 python tools/mt_batch_digits_time.py 24 32 | tail -n 3
timeout 300 python tools/mt_many_time.py 8 32 | tail -n 2
timeout 300 python tools/mt_many_time.py 32 32 | tail -n 2
timeout 300 python tools/mt_many_time.py 8 1 | tail -n 1
} > gpurun_out/r05_mt_many4.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch_c4.py tests/test_gpu_crypto.py -m gpu -x -q 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP" | tail -n 5 >> gpurun_out/r05_mt_many4.txt
timeout 400 python tests/manual/stress_gangs.py 300 11 >> gpurun_out/r05_mt_many4.txt 2>&1
