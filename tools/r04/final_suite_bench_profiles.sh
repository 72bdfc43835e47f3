#!/bin/bash
# round 4: full suite, smoke, default bench line, profiles of the final build
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q > $O/r04_pytest12.log 2>&1; echo "full suite rc=$?" > $O/r04_gpu12.summary
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/r04_smoke12.log 2>&1; echo "smoke rc=$?" >> $O/r04_gpu12.summary
python bench.py > $O/r04_bench_default.json 2> $O/r04_bench_default.err; echo "bench rc=$?" >> $O/r04_gpu12.summary
bash tools/jobs/kernel_stats.sh r04_bench python bench.py --no-cpu-baseline --no-batch-c4 --no-extra-legs --target-n 0
bash tools/jobs/kernel_stats.sh r04_65536 python tools/profile_one.py 65536 1
bash tools/jobs/kernel_stats.sh r04_262144 python tools/profile_one.py 262144 1
KEEP_TRACE=1 bash tools/jobs/kernel_stats.sh r04_batch python tools/profile_batch.py 32768 128 2
python tools/gang_budget.py $O/r04_batch_trace 128 > $O/r04_batch_budget_final.txt 2>&1
find $O/r04_batch_trace -name "*kernel_trace.csv" -delete
bash tools/jobs/pmc_traffic.sh r04_262144_k16k "k_update16k" --range "[1-6]" -- python tools/profile_one.py 262144 1
bash tools/jobs/pmc_traffic.sh r04_65536 "k_update16<" -- python tools/profile_one.py 65536 1
