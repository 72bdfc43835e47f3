#!/bin/bash
# round 5, the closing lease: full suite, smoke, default bench line, open-ended differential runs, profiles of the final build
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 2700 python -m pytest tests -m gpu -q > $O/r05_pytest_final.log 2>&1; echo "full suite rc=$?" > $O/r05_final.summary
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/r05_smoke_final.log 2>&1; echo "smoke rc=$?" >> $O/r05_final.summary
python bench.py > $O/r05_bench_final.json 2> $O/r05_bench_final.err; echo "bench rc=$?" >> $O/r05_final.summary
timeout 400 python tests/manual/stress_parity.py 240 515 > $O/r05_stress_final.log 2>&1; echo "stress rc=$?" >> $O/r05_final.summary
timeout 300 python tests/manual/stress_gangs.py 150 7 > $O/r05_stress_gangs_final.log 2>&1; echo "stress gangs rc=$?" >> $O/r05_final.summary
if [ -n "$WITH_PROFILES" ]; then
  bash tools/jobs/kernel_stats.sh r05_bench python bench.py --no-cpu-baseline --no-batch-c4 --no-extra-legs --target-n 0
  bash tools/jobs/kernel_stats.sh r05_65536 python tools/profile_one.py 65536 1
  SEED=1242 bash tools/jobs/kernel_stats.sh r05_262144 python tools/profile_one.py 262144 1
  bash tools/jobs/kernel_stats.sh r05_batch python tools/profile_batch.py 32768 128 2
  for bs in 32 17 1; do bash tools/jobs/kernel_stats.sh r05_c3_mt$bs python tools/mt_stats.py $bs; done
  bash tools/jobs/kernel_stats.sh r05_c5_xoshiro python examples/xoshiro_recovery.py
  SEED=1242 bash tools/jobs/pmc_traffic.sh r05_262144_k16k "k_update16k" --range "[1-6]" -- python tools/profile_one.py 262144 1
  SEED=1242 bash tools/jobs/pmc_traffic.sh r05_262144_k16_inner "k_update16<" --range "[1-48]" -- python tools/profile_one.py 262144 1
  bash tools/jobs/pmc_traffic.sh r05_65536 "k_update16<" -- python tools/profile_one.py 65536 1
  GF2BV_GANG=24 bash tools/jobs/pmc_traffic.sh r05_gang24 "k_update16<" -- python tools/profile_batch.py 32768 24 1
fi
