#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd $R
{ for rpt in 1 2 4 1 2 4; do echo "## GF2BV_FUSED_RPT=$rpt"; for n in 32768 65536; do GF2BV_FUSED_RPT=$rpt timeout 120 python tools/profile_one.py $n 4 | tail -2; done; done
  echo "## default"; for n in 16384 32768 65536 131072; do timeout 120 python tools/profile_one.py $n 3 | tail -1; done; } > $O/r03_rpt44.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/r03_trace_65536b -- python $R/tools/profile_one.py 65536 2 > $O/r03_trace_65536b.log 2>&1
cd $R; python tools/pass_rates.py $O/r03_trace_65536b 65536 1 > $O/r03_pass_rates_b.txt 2>&1
find $O/r03_trace_65536b -name "*.csv" ! -name "*kernel_trace.csv" -delete; gzip -f $(find $O/r03_trace_65536b -name "*kernel_trace.csv")
