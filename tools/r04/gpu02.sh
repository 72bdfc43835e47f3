#!/bin/bash
# round 4: gang-wide back-substitution + staggered gangs: parity, then the A/B on 192 x 32768^2 (8 gangs of 24)
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch_c4.py tests/test_gpu_slab.py -x -q -k "gang or batch or c4 or slab or devices" > $O/r04_pytest02.log 2>&1; echo "pytest rc=$?" > $O/r04_gpu02.summary
{
for cfg in "" "GF2BV_STAGGER=0" "GF2BV_GANG_BS=0" "GF2BV_STAGGER=0 GF2BV_GANG_BS=0" "GF2BV_BATCH_THREADS=3" "GF2BV_BATCH_THREADS=4" "GF2BV_BATCH_THREADS=1" \
           "GF2BV_GANG=12" "GF2BV_GANG=16" "GF2BV_GANG=32" "GF2BV_GANG=16 GF2BV_BATCH_THREADS=3" "GF2BV_GANG=12 GF2BV_BATCH_THREADS=4" "GF2BV_GANG=8 GF2BV_BATCH_THREADS=4"; do
  echo "## $cfg"; env $cfg timeout 300 python tools/batch_time.py 32768 192 | grep batch
done
} > $O/r04_batch_ab02.txt 2>&1
