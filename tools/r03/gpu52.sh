#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd $R
for g in "" 4 8 12 18 24 36; do echo "## GF2BV_GANG=$g"; GF2BV_GANG=$g python bench.py --workload batch --batch-total 144 --no-cpu-baseline --steps 1 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d.get('systems_per_s'), d['ms_per_step'], d['roofline']['frac'] if d.get('roofline') else None, d['config']['parallelism'])"; done > $O/r03_gang52.txt 2>&1
