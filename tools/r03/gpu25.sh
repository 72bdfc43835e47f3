#!/bin/bash
# kernel traces of two-level solves: where the per-panel overhead goes (tools/outer_timeline.py)
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
GF2BV_TWO_LEVEL=8 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/r03_tl_65536_k8 -- python $R/tools/profile_one.py 65536 1 > $O/r03_tl_65536_k8.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/r03_tl_131072 -- python $R/tools/profile_one.py 131072 1 > $O/r03_tl_131072.log 2>&1
cd $R
python tools/outer_timeline.py $O/r03_tl_65536_k8 2 > $O/r03_outer_timeline.txt 2>&1
python tools/outer_timeline.py $O/r03_tl_131072 4 >> $O/r03_outer_timeline.txt 2>&1
find $O/r03_tl_65536_k8 $O/r03_tl_131072 -name "*.csv" ! -name "*kernel_trace.csv" -delete
gzip -f $(find $O/r03_tl_65536_k8 $O/r03_tl_131072 -name "*kernel_trace.csv")
