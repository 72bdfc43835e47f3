#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd $R
{
for cfg in "" "GF2BV_BATCH_THREADS=3" "GF2BV_XCD_WGS=24" "GF2BV_BATCH_THREADS=3 GF2BV_XCD_WGS=24" "GF2BV_GANG=64" ""; do
  echo "## $cfg"; env $cfg timeout 300 python bench.py --workload batch --steps 3 --warmup 1 --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['systems_per_s'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['end_to_end_frac'], d['config']['parallelism'])"
done
} > $O/r04_batch_bench15.txt 2>&1
