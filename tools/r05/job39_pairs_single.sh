#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
{
for r in 1 2 3; do echo "== run $r"; GF2BV_TRACE=1 timeout 300 python tools/mt_batch_digits_time.py 16 32 2>&1 | grep "stream pair\|one by one\|call" | cut -c1-140; done
echo "== singles first, then batches"; GF2BV_TRACE=1 timeout 300 python tools/mt_many_time.py 8 32 2>&1 | grep "stream pair\|MT19937" | cut -c1-200
echo "== batch 32768 x 64"; timeout 600 python tools/batch_time.py 32768 64 2 2>&1 | tail -n 1
echo "== 65536"; timeout 300 python tools/profile_one.py 65536 3 2>&1 | grep "N=" | cut -c1-100
} > gpurun_out/r05_pairs_single.txt
