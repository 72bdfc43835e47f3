#!/bin/bash
# round 4, the closing lease: full suite, smoke, default bench line, open-ended differential run, profiles of the final build
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q > $O/r04_pytest_final.log 2>&1; echo "full suite rc=$?" > $O/r04_final.summary
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/r04_smoke_final.log 2>&1; echo "smoke rc=$?" >> $O/r04_final.summary
python bench.py > $O/r04_bench_final.json 2> $O/r04_bench_final.err; echo "bench rc=$?" >> $O/r04_final.summary
timeout 500 python tests/manual/stress_parity.py 300 505 > $O/r04_stress_final.log 2>&1; echo "stress rc=$?" >> $O/r04_final.summary
if [ -n "$WITH_PROFILES" ]; then
  bash tools/jobs/kernel_stats.sh r04_bench python bench.py --no-cpu-baseline --no-batch-c4 --no-extra-legs --target-n 0
  bash tools/jobs/kernel_stats.sh r04_65536 python tools/profile_one.py 65536 1
  bash tools/jobs/kernel_stats.sh r04_262144 python tools/profile_one.py 262144 1
  bash tools/jobs/kernel_stats.sh r04_batch python tools/profile_batch.py 32768 128 2
  bash tools/jobs/pmc_traffic.sh r04_262144_k16k "k_update16k" --range "[1-6]" -- python tools/profile_one.py 262144 1
  bash tools/jobs/pmc_traffic.sh r04_65536 "k_update16<" -- python tools/profile_one.py 65536 1
  GF2BV_GANG=24 bash tools/jobs/pmc_traffic.sh r04_gang24 "k_update16<" -- python tools/profile_batch.py 32768 24 1
fi
