#!/bin/bash
# Round 6: where k_block_sparse spends its time on the two MT19937 variants that miss their targets (9 bits / 1 bit per output),
# and a kernel timeline of the one-bit solve; + RCCL init A/B on this (fresh) lease.
mkdir -p gpurun_out
python tools/r06/rccl_init_ab.py "lease $(date +%H%M)" >> gpurun_out/r06_rccl_init.txt 2>&1
python tools/probe_sparse.py 32 9 1 2>&1 | grep -v "gives up" | tail -12 > gpurun_out/r06e_sparse_probe.txt
cat gpurun_out/r06e_sparse_probe.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_mt1 -- python $GRAFT_REPO_ROOT/tools/mt_stats.py 1 > /tmp/prof_mt1.log 2>&1
python - <<'PY' > $GRAFT_REPO_ROOT/gpurun_out/r06e_mt1_kernel_stats.txt
import csv, glob, collections
f = glob.glob('/tmp/prof_mt1/**/*kernel_stats.csv', recursive=True)
for path in f:
    for row in csv.DictReader(open(path)):
        print(f"{int(row['Calls']):8d} {float(row['TotalDurationNs'])/1e3:12.1f} {float(row['AverageNs'])/1e3:10.2f} {float(row['Percentage']):6.2f}  {row['Name'][:90]}")
PY
cat $GRAFT_REPO_ROOT/gpurun_out/r06e_mt1_kernel_stats.txt | head -20
