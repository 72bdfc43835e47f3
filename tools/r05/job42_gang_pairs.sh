#!/bin/bash
# gang stream sets chosen by the chain probe (default) against whatever the pool hands out (GF2BV_GANG_PAIRS=0)
cd /root/repo; mkdir -p gpurun_out
{
for r in 1 2; do
echo "== MT 16, gang pairs on, run $r"; GF2BV_TRACE=1 timeout 300 python tools/mt_batch_digits_time.py 16 32 2>&1 | grep "stream pair\|one by one\|call" | cut -c1-130
echo "== MT 16, gang pairs off, run $r"; GF2BV_GANG_PAIRS=0 timeout 300 python tools/mt_batch_digits_time.py 16 32 2>&1 | grep "one by one\|call" | cut -c1-110
done
for r in 1 2; do
echo "== batch 32768 x 64 on"; timeout 600 python tools/batch_time.py 32768 64 2 2>&1 | tail -n 1
echo "== batch 32768 x 64 off"; GF2BV_GANG_PAIRS=0 timeout 600 python tools/batch_time.py 32768 64 2 2>&1 | tail -n 1
done
echo "== batch 32768 x 192 on"; timeout 600 python tools/batch_time.py 32768 192 2 2>&1 | tail -n 1
echo "== batch 32768 x 192 off"; GF2BV_GANG_PAIRS=0 timeout 600 python tools/batch_time.py 32768 192 2 2>&1 | tail -n 1
echo "== 65536 / 262144 (chain probe for sB and sC)"; GF2BV_TRACE=1 timeout 300 python tools/profile_one.py 65536 3 2>&1 | grep "stream pair\|N=" | cut -c1-110
SEED=1242 GF2BV_TRACE=1 timeout 600 python tools/profile_one.py 262144 2 2>&1 | grep "stream pair\|N=" | cut -c1-110
} > gpurun_out/r05_gang_pairs.txt
