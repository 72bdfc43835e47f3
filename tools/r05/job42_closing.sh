#!/bin/bash
# late round 5, closing: full GPU suite, smoke, the default bench line, kernel stats of the bench command and of the 262144^2 leg (final build: one outer stream)
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
timeout 1700 python -m pytest tests -m gpu -q > $O/r05y_pytest.log 2>&1; echo "full suite rc=$?" > $O/r05y.summary
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/r05y_smoke.log 2>&1; echo "smoke rc=$?" >> $O/r05y.summary
python bench.py > $O/r05y_bench_default.json 2> $O/r05y_bench_default.err; echo "bench rc=$?" >> $O/r05y.summary
bash tools/jobs/kernel_stats.sh r05y_bench python bench.py --no-cpu-baseline --no-batch-c4 --no-extra-legs --target-n 0
SEED=1242 bash tools/jobs/kernel_stats.sh r05y_262144 python tools/profile_one.py 262144 1
SEED=1242 bash tools/jobs/pmc_traffic.sh r05y_262144_k16k "k_update16k" --range "[1-6]" -- python tools/profile_one.py 262144 1
cat $O/r05y.summary; tail -2 $O/r05y_pytest.log
