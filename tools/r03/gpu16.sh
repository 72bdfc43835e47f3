#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_stress.py -x -q -k two_level > $O/r03_pytest16a.log 2>&1; echo "stress two-level rc=$?" > $O/r03_pytest16.summary
GF2BV_TWO_LEVEL=8 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q > $O/r03_pytest16b.log 2>&1; echo "parity K=8 rc=$?" >> $O/r03_pytest16.summary
GF2BV_TWO_LEVEL=3 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q > $O/r03_pytest16c.log 2>&1; echo "parity K=3 rc=$?" >> $O/r03_pytest16.summary
{ for tl in "" 8; do echo "## GF2BV_TWO_LEVEL=$tl"; GF2BV_TWO_LEVEL=$tl timeout 120 python tools/profile_one.py 131072 3 | tail -2; GF2BV_TWO_LEVEL=$tl timeout 200 python tools/profile_one.py 262144 3 | tail -2; done
  echo "## 65536 / 98304 forced 8"; GF2BV_TWO_LEVEL=8 python tools/profile_one.py 65536 3 | tail -1; GF2BV_TWO_LEVEL=8 python tools/profile_one.py 98304 3 | tail -1; GF2BV_TWO_LEVEL=0 python tools/profile_one.py 98304 3 | tail -1; } > $O/r03_two_level_times8.txt 2>&1
cd /tmp; export TMPDIR=/tmp
GF2BV_TWO_LEVEL=8 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r03_tl8d_131072 -- python $R/tools/profile_one.py 131072 1 > $O/r03_tl8d_131072.log 2>&1
