#!/bin/bash
# Round 6: the whole GPU suite on the pruned build (driver's form) + the default bench line.
mkdir -p gpurun_out
( time python -m pytest tests/ -x -q -m gpu --durations=12 ) > gpurun_out/r06c_pytest.log 2>&1
tail -22 gpurun_out/r06c_pytest.log
python bench.py > gpurun_out/r06c_bench.json 2> gpurun_out/r06c_bench.err
tail -5 gpurun_out/r06c_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r06c_bench.json'))
print({k:d.get(k) for k in ('ms_per_step','plain_ms_per_step','build_id')}, d['parity_gate'])
t=d['target_262144']; print('target', t['ms_per_step'], t['roofline'].get('elimination_frac'), t['roofline'].get('hbm_real_frac'), t['x_equals_planted'])
print('c4', d['batch_c4']['systems_per_s'])
for v in d['c3_mt19937']['variants']: print(v['bits_per_output'], v['device_ms']['eliminate'], v['m4ri_solve_ms']['warm'])
PY
