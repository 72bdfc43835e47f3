#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd $R
for f in 1 0 1 0; do echo "## GF2BV_FUSED_NARROW=$f"; GF2BV_FUSED_NARROW=$f python bench.py --workload batch --batch-total 144 --no-cpu-baseline --steps 1 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d.get('systems_per_s'), d['ms_per_step'], d['roofline']['frac'] if d.get('roofline') else None)"; done > $O/r03_fused31.txt 2>&1
{ for f in 1 0; do echo "## GF2BV_FUSED_NARROW=$f MT / small"; GF2BV_FUSED_NARROW=$f timeout 300 python examples/mt_recovery.py 2>&1 | tail -4; done; } >> $O/r03_fused31.txt 2>&1
