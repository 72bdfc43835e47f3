#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
{
for r in 1 2 3; do echo "== run $r"; GF2BV_TRACE=1 timeout 300 python tools/mt_batch_digits_time.py 16 32 2>&1 | grep "stream pair\|one by one\|call" | cut -c1-140; done
echo "== pairs off"; GF2BV_STREAM_PAIRS=0 timeout 300 python tools/mt_batch_digits_time.py 16 32 2>&1 | grep "one by one" | cut -c1-100
echo "== 65536"; GF2BV_TRACE=1 timeout 300 python tools/profile_one.py 65536 3 2>&1 | grep "stream pair\|N=" | cut -c1-140
echo "== 262144"; SEED=1242 GF2BV_TRACE=1 timeout 600 python tools/profile_one.py 262144 2 2>&1 | grep "stream pair\|N=" | cut -c1-140
echo "== batch 32768 x 64"; timeout 600 python tools/batch_time.py 32768 64 2 2>&1 | tail -n 3
} > gpurun_out/r05_pairs.txt
