#!/bin/bash
# Round 6 closing lease: RCCL init A/B on this fresh box, the GPU suite in the driver's form, smoke, the default bench line, kernel stats of the bench command /
# the 262144^2 leg / a batch leg, PMC traffic of the two dominant kernels, open-ended differential runs.
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
python tools/r06/rccl_init_ab.py "lease $(date +%H%M)" >> $O/r06_rccl_init.txt 2>&1
( time timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=8 ) > $O/r06z_pytest.log 2>&1; echo "full suite rc=$?" > $O/r06z.summary
python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" > $O/r06z_smoke.log 2>&1; echo "smoke rc=$?" >> $O/r06z.summary
python bench.py > $O/r06z_bench_default.json 2> $O/r06z_bench_default.err; echo "bench rc=$?" >> $O/r06z.summary
bash tools/jobs/kernel_stats.sh r06z_bench python bench.py --no-cpu-baseline --no-batch-c4 --no-extra-legs --no-plain-leg --target-n 0
SEED=1242 bash tools/jobs/kernel_stats.sh r06z_262144 python tools/profile_one.py 262144 1
bash tools/jobs/kernel_stats.sh r06z_batch python tools/profile_batch.py 32768 64 1
bash tools/jobs/pmc_traffic.sh r06z_65536 "k_update16<" -- python tools/profile_one.py 65536 1
SEED=1242 bash tools/jobs/pmc_traffic.sh r06z_262144_k16k "k_update16k" --range "[1-6]" -- python tools/profile_one.py 262144 1
timeout 1000 python tests/manual/stress_parity.py 900 7007 > $O/r06z_stress_parity.log 2>&1; echo "stress_parity rc=$?" >> $O/r06z.summary
timeout 420 python tests/manual/stress_gangs.py 300 707 > $O/r06z_stress_gangs.log 2>&1; echo "stress_gangs rc=$?" >> $O/r06z.summary
cat $O/r06z.summary; tail -3 $O/r06z_pytest.log; tail -2 $O/r06z_stress_parity.log; tail -2 $O/r06z_stress_gangs.log; tail -9 $O/r06_rccl_init.txt
