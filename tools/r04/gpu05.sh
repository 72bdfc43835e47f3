#!/bin/bash
# round 4: a gang's bulk update, one system after the other on its XCD; WG count per system; non-temporal row loads / stores
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd $R
{
for cfg in "GF2BV_XCD_PIN=1" "GF2BV_XCD_PIN=2" "GF2BV_XCD_PIN=0" "GF2BV_XCD_PIN=1 GF2BV_XCD_WGS=16" "GF2BV_XCD_PIN=1 GF2BV_XCD_WGS=24" "GF2BV_XCD_PIN=1 GF2BV_XCD_WGS=64" \
           "GF2BV_XCD_PIN=1 GF2BV_GANG=16" "GF2BV_XCD_PIN=1 GF2BV_GANG=32" "GF2BV_XCD_PIN=1 GF2BV_GANG=48" "GF2BV_XCD_PIN=1 GF2BV_BATCH_THREADS=1" "GF2BV_XCD_PIN=1 GF2BV_BATCH_THREADS=3" \
           "GF2BV_XCD_PIN=1 GF2BV_LIB=$R/tools/_probe/lib_NT_LOAD.so" "GF2BV_XCD_PIN=1 GF2BV_LIB=$R/tools/_probe/lib_NT_STORE.so" "GF2BV_XCD_PIN=1 GF2BV_LIB=$R/tools/_probe/lib_NT_BOTH.so" \
           "GF2BV_XCD_PIN=2 GF2BV_LIB=$R/tools/_probe/lib_NT_LOAD.so" "GF2BV_XCD_PIN=2 GF2BV_LIB=$R/tools/_probe/lib_NT_BOTH.so" "GF2BV_XCD_PIN=1"; do
  echo "## $cfg"; env $cfg timeout 300 python tools/batch_time.py 32768 192 5 | grep batch
done
} > $O/r04_batch_ab05.txt 2>&1
GF2BV_GANG=24 GF2BV_XCD_PIN=1 bash tools/jobs/pmc_traffic.sh r04_gang24_seq "k_update16<" -- python tools/profile_batch.py 32768 24 1
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "gang or batch or update_conf" > $O/r04_pytest05.log 2>&1; echo "rc=$?" > $O/r04_gpu05.summary
