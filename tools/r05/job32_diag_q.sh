#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
{
echo "== GPU_MAX_HW_QUEUES=8"; GPU_MAX_HW_QUEUES=8 timeout 300 python tools/mt_batch_digits_time.py 16 32 | cut -c1-200
echo "== GPU_MAX_HW_QUEUES=2"; GPU_MAX_HW_QUEUES=2 timeout 300 python tools/mt_batch_digits_time.py 16 32 | cut -c1-200
echo "== default"; timeout 300 python tools/mt_batch_digits_time.py 16 32 | cut -c1-200
echo "== default, 5 reps of singles only after one batch"; 
} > gpurun_out/r05_diag_q.txt 2>&1
