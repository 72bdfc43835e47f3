#!/bin/bash
# round 4: XCD pinning of a gang's bulk update (A/B + PMC), the gang two-level trace (why it loses), new k_to_tiled, full GPU suite
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd $R
{
for cfg in "GF2BV_XCD_PIN=1" "GF2BV_XCD_PIN=0" "GF2BV_XCD_PIN=1 GF2BV_GANG=16" "GF2BV_XCD_PIN=0 GF2BV_GANG=16" "GF2BV_XCD_PIN=1 GF2BV_GANG=32" "GF2BV_XCD_PIN=0 GF2BV_GANG=32" \
           "GF2BV_XCD_PIN=1 GF2BV_BATCH_THREADS=1" "GF2BV_XCD_PIN=0 GF2BV_BATCH_THREADS=1" "GF2BV_XCD_PIN=1 GF2BV_GANG=16 GF2BV_BATCH_THREADS=3" "GF2BV_XCD_PIN=1 GF2BV_STAGGER=0" \
           "GF2BV_XCD_PIN=1" "GF2BV_XCD_PIN=0"; do
  echo "## $cfg"; env $cfg timeout 300 python tools/batch_time.py 32768 192 5 | grep batch
done
} > $O/r04_batch_ab04.txt 2>&1
GF2BV_GANG=24 GF2BV_XCD_PIN=1 bash tools/jobs/pmc_traffic.sh r04_gang24_pin "k_update16<" -- python tools/profile_batch.py 32768 24 1
GF2BV_GANG=24 GF2BV_XCD_PIN=0 bash tools/jobs/pmc_traffic.sh r04_gang24_nopin "k_update16<" -- python tools/profile_batch.py 32768 24 1
GF2BV_GANG_TWO_LEVEL=1 KEEP_TRACE=1 bash tools/jobs/kernel_stats.sh r04_batch_tl python tools/profile_batch.py 32768 48 2
python tools/gang_budget.py $O/r04_batch_tl_trace 48 > $O/r04_batch_tl_budget.txt 2>&1
KEEP_TRACE=1 bash tools/jobs/kernel_stats.sh r04_batch_1l python tools/profile_batch.py 32768 96 2
python tools/gang_budget.py $O/r04_batch_1l_trace 96 > $O/r04_batch_1l_budget.txt 2>&1
bash tools/jobs/kernel_stats.sh r04_262144 python tools/profile_one.py 262144 2
timeout 900 python -m pytest tests/test_gpu_batch_c4.py tests/test_gpu_slab.py -x -q > $O/r04_pytest04b.log 2>&1; echo "c4+slab rc=$?" > $O/r04_gpu04.summary
