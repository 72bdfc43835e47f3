#!/bin/bash
O=gpurun_out; mkdir -p $O
{
echo "# 512 x 32768^2 on one GPU, tools/batch_time.py, by host threads / gang size / workgroups per system"
for t in 2 3 4; do echo "## GF2BV_BATCH_THREADS=$t"; GF2BV_BATCH_THREADS=$t timeout 600 python tools/batch_time.py 32768 512 3; done
for g in 16 24 40 48 64; do echo "## GF2BV_GANG=$g"; GF2BV_GANG=$g timeout 600 python tools/batch_time.py 32768 512 3; done
for w in 24 28 36 40; do echo "## GF2BV_XCD_WGS=$w"; GF2BV_XCD_WGS=$w timeout 600 python tools/batch_time.py 32768 512 3; done
echo "## threads 3 + gang 24"; GF2BV_BATCH_THREADS=3 GF2BV_GANG=24 timeout 600 python tools/batch_time.py 32768 512 3
} > $O/r05_batch_scan2.txt 2>&1
