#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/r03_trace_65536 -- python $R/tools/profile_one.py 65536 2 > $O/r03_trace_65536.log 2>&1
cd $R; python tools/pass_rates.py $O/r03_trace_65536 65536 8 > $O/r03_pass_rates.txt 2>&1
find $O/r03_trace_65536 -name "*.csv" ! -name "*kernel_trace.csv" -delete; gzip -f $(find $O/r03_trace_65536 -name "*kernel_trace.csv")
