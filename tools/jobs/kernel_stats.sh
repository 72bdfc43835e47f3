#!/bin/bash
# rocprofv3 --kernel-trace --stats of one command -> gpurun_out/<tag>_kernel_stats.txt (the table committed under profiles/)
# usage (on the GPU box, from the repo root):  bash tools/jobs/kernel_stats.sh <tag> <command ...>
#   e.g. bash tools/jobs/kernel_stats.sh r04_65536 python tools/profile_one.py 65536 1
# KEEP_TRACE=1 keeps the per-dispatch csv (tools/outer_timeline.py, tools/b_idle.py, tools/gang_budget.py read it).
tag=$1; shift
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
args=(); for a in "$@"; do if [ -e "$R/$a" ]; then args+=("$R/$a"); else args+=("$a"); fi; done; set -- "${args[@]}"      # (the command runs in /tmp)
cd /tmp && export TMPDIR=/tmp
timeout ${JOB_TIMEOUT:-900} rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_trace -- "$@" > $O/${tag}_trace.log 2>&1
cd $R
python tools/rocprof_csv_summary.py $O/${tag}_trace "$*" > $O/${tag}_kernel_stats.txt 2>&1
[ -n "$KEEP_TRACE" ] || find $O/${tag}_trace -name "*kernel_trace.csv" -delete
find $O/${tag}_trace -name "*kernel_trace.csv" -size +40M -delete
