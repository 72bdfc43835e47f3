#!/bin/bash
# HBM-side traffic of one kernel family (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes, no trace
# domains besides --kernel-trace) -> gpurun_out/<tag>_pmc.txt
# usage: bash tools/jobs/pmc_traffic.sh <tag> <kernel regex> [--range "[1-32]"] -- <command ...>
# Under --pmc kernels of different streams do not run concurrently: the solver detects that and hands over through events.
tag=$1; regex=$2; shift 2
range=()
if [ "$1" == "--range" ]; then range=(--kernel-iteration-range "$2"); shift 2; fi
[ "$1" == "--" ] && shift
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
args=(); for a in "$@"; do if [ -e "$R/$a" ]; then args+=("$R/$a"); else args+=("$a"); fi; done; set -- "${args[@]}"      # (the command runs in /tmp)
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout ${JOB_TIMEOUT:-900} rocprofv3 --pmc $c --kernel-include-regex "$regex" "${range[@]}" --kernel-trace --output-format csv -d $O/${tag}_$c -- "$@" > $O/${tag}_$c.log 2>&1
done
cd $R
{ echo "# $* ; kernels matching '$regex' ${range[*]}"; for c in FETCH_SIZE WRITE_SIZE; do python tools/pmc_summary.py $O/${tag}_$c; done; } > $O/${tag}_pmc.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do find $O/${tag}_$c -name "*.csv" -delete; done
