#!/bin/bash
# round 4: the 512-system job: gang size, event brackets, threads
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd $R
{
for cfg in "" "TIME_KERNELS=1" "GF2BV_GANG=24" "GF2BV_GANG=24 TIME_KERNELS=1" "GF2BV_BATCH_THREADS=3" "GF2BV_BATCH_THREADS=3 TIME_KERNELS=1" "GF2BV_GANG=64" "GF2BV_GANG=16 GF2BV_BATCH_THREADS=4" ; do
  echo "## $cfg"; env $cfg timeout 300 python tools/batch_time.py 32768 512 4 | grep batch
done
} > $O/r04_batch_ab07.txt 2>&1
