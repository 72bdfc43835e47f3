#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd $R
{ for tl in "" 4 8; do echo "## GF2BV_TWO_LEVEL=$tl"; for n in 131072; do GF2BV_TWO_LEVEL=$tl timeout 120 python tools/profile_one.py $n 3 | tail -2; done; GF2BV_TWO_LEVEL=$tl timeout 200 python tools/profile_one.py 262144 3 | tail -2; done; } > $O/r03_two_level_times3.txt 2>&1
GF2BV_TWO_LEVEL=3 timeout 600 python -m pytest tests/test_gpu_stress.py -x -q -k two_level > $O/r03_pytest8.log 2>&1; echo "rc=$?" >> $O/r03_pytest8.log
cd /tmp; export TMPDIR=/tmp
GF2BV_TWO_LEVEL=4 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r03_tl4b_131072 -- python $R/tools/profile_one.py 131072 1 > $O/r03_tl4b_131072.log 2>&1
