#!/bin/bash
# round 4: streaming row accesses for pinned gangs (template instance), parity of the pinned path, full bench line
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "gang or batch" > $O/r04_pytest06.log 2>&1; echo "rc=$?" > $O/r04_gpu06.summary
{
for cfg in "" "GF2BV_GANG_NT=0" "GF2BV_BATCH_THREADS=3" "GF2BV_GANG=32" "GF2BV_XCD_WGS=24" ""; do
  echo "## $cfg"; env $cfg timeout 300 python tools/batch_time.py 32768 192 5 | grep batch
done
} > $O/r04_batch_ab06.txt 2>&1
python bench.py > $O/r04_bench_a.json 2> $O/r04_bench_a.err; echo "bench rc=$?" >> $O/r04_gpu06.summary
