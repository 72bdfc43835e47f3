#!/bin/bash
# gangs on the two-launch flow again: batch tests, stress, batch throughput, whole suite
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -m gpu -x -q > $O/r03_pytest53.log 2>&1; echo "full suite rc=$?" > $O/r03_final53.summary
for g in "" 24; do echo "## GF2BV_GANG=$g"; GF2BV_GANG=$g python bench.py --workload batch --batch-total 144 --no-cpu-baseline --steps 1 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d.get('systems_per_s'), d['ms_per_step'], d['config']['parallelism'])"; done > $O/r03_gang53.txt 2>&1
