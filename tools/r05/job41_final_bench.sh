#!/bin/bash
# late round 5, closing: smoke, the default bench line (ms_sweep = union of the bulk launches), kernel stats of the bench command, differential runs
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/r05x_smoke.log 2>&1; echo "smoke rc=$?" > $O/r05x.summary
python bench.py > $O/r05x_bench_default.json 2> $O/r05x_bench_default.err; echo "bench rc=$?" >> $O/r05x.summary
bash tools/jobs/kernel_stats.sh r05x_bench python bench.py --no-cpu-baseline --no-batch-c4 --no-extra-legs --target-n 0
timeout 400 python tests/manual/stress_parity.py 200 616 > $O/r05x_stress.log 2>&1; echo "stress rc=$?" >> $O/r05x.summary
cat $O/r05x.summary; tail -3 $O/r05x_stress.log
