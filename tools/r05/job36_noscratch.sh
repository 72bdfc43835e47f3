#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
{
echo "== run 1"; timeout 300 python tools/mt_batch_digits_time.py 16 32 2>&1 | cut -c1-130
echo "== run 2"; timeout 300 python tools/mt_batch_digits_time.py 16 32 2>&1 | cut -c1-130
echo "== run 3"; timeout 300 python tools/mt_batch_digits_time.py 16 32 2>&1 | cut -c1-130
echo "== 65536"; timeout 300 python tools/profile_one.py 65536 3 2>&1 | tail -n 4
echo "== 32768"; timeout 300 python tools/profile_one.py 32768 3 2>&1 | tail -n 3
} > gpurun_out/r05_diag_noscratch.txt
