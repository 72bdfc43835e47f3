#!/bin/bash
# box variance of the bench line: the default bench (no CPU baseline, no example legs) on a fresh box
mkdir -p gpurun_out
python bench.py --no-cpu-baseline --no-extra-legs > gpurun_out/r06_box_$(date +%H%M).json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_box_*.json')):
    d=json.load(open(f)); t=d['target_262144']
    print(f, round(d['ms_per_step'],2), round(d['plain_ms_per_step'],2), round(d['roofline']['frac'],4), round(t['solve_wall_ms']['eliminate'],1), round(t['roofline']['elimination_frac'],4), round(d['batch_c4']['systems_per_s'],1))
PY
