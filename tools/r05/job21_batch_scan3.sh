#!/bin/bash
O=gpurun_out; mkdir -p $O
{
for c in "3 64" "4 64" "3 48" "3 32" "2 64" "3 128"; do set -- $c; echo "## threads $1 gang $2"; GF2BV_BATCH_THREADS=$1 GF2BV_GANG=$2 timeout 600 python tools/batch_time.py 32768 512 4; done
echo "## 64 systems (one rank's share at 8 GPUs): threads 2 / 3, gang default / 32"
for c in "2 0" "3 0" "2 32" "3 32"; do set -- $c; echo "## 64 systems: threads $1 gang $2"; if [ $2 == 0 ]; then GF2BV_BATCH_THREADS=$1 timeout 600 python tools/batch_time.py 32768 64 4; else GF2BV_BATCH_THREADS=$1 GF2BV_GANG=$2 timeout 600 python tools/batch_time.py 32768 64 4; fi; done
} > $O/r05_batch_scan3.txt 2>&1
