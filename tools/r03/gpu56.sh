#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_batch_c4.py tests/test_gpu_stress.py -x -q > $O/r03_pytest56.log 2>&1; echo "batch+stress tests rc=$?" > $O/r03_final56.summary
python bench.py --no-cpu-baseline --target-n 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d['batch_c4']; print('512 systems:', b['systems_per_s'], b['ms_per_step'], b['config']['parallelism'], b['roofline']['end_to_end_frac'])" > $O/r03_gang56.txt 2>&1
for t in 64 144; do python bench.py --workload batch --batch-total $t --no-cpu-baseline --steps 1 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d.get('systems_per_s'), d['ms_per_step'], d['config']['parallelism'])"; done >> $O/r03_gang56.txt 2>&1
