#!/bin/bash
# round 4: two-level elimination of gangs: parity, then scans on 192 x 32768^2
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_stress.py tests/test_gpu_parity.py tests/test_gpu_batch_c4.py -x -q -k "gang or batch or c4 or two_level" > $O/r04_pytest03.log 2>&1; echo "pytest rc=$?" > $O/r04_gpu03.summary
{
for cfg in "" "GF2BV_GANG_TWO_LEVEL=0" "GF2BV_TWO_LEVEL_MIN_MIB=128" "GF2BV_TWO_LEVEL_MIN_MIB=256" "GF2BV_TWO_LEVEL_MIN_MIB=1024" "GF2BV_TWO_LEVEL_MIN_MIB=2048" \
           "GF2BV_OUTER_K=4" "GF2BV_OUTER_K=6" "GF2BV_OUTER_K=12" \
           "GF2BV_BATCH_THREADS=1" "GF2BV_BATCH_THREADS=3" "GF2BV_BATCH_THREADS=4" "GF2BV_STAGGER=0" \
           "GF2BV_GANG=12" "GF2BV_GANG=16" "GF2BV_GANG=32" "GF2BV_GANG=48" "GF2BV_GANG=12 GF2BV_BATCH_THREADS=4" "GF2BV_GANG=16 GF2BV_BATCH_THREADS=3" \
           "GF2BV_GANG=48 GF2BV_BATCH_THREADS=1" "GF2BV_GANG=32 GF2BV_BATCH_THREADS=3"; do
  echo "## $cfg"; env $cfg timeout 300 python tools/batch_time.py 32768 192 | grep batch
done
} > $O/r04_batch_ab03.txt 2>&1
KEEP_TRACE=1 bash tools/jobs/kernel_stats.sh r04_batch_tl python tools/profile_batch.py 32768 96 2
python tools/gang_budget.py $O/r04_batch_tl_trace 96 > $O/r04_batch_tl_budget.txt 2>&1
