#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd /tmp
{ for b in mbk2_s16 mbk2_s20 mbk2_s24; do $R/tools/_probe/$b 131072 1024 4; $R/tools/_probe/$b 131072 1024 8; MB_WGS=240 $R/tools/_probe/$b 131072 1024 4; done; } > $O/r03_kloop6.txt 2>&1
cd $R
GF2BV_TWO_LEVEL=2 timeout 600 python -m pytest tests/test_gpu_stress.py -x -q -k two_level > $O/r03_pytest12a.log 2>&1; echo "stress rc=$?" > $O/r03_pytest12.summary
GF2BV_TWO_LEVEL=4 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q > $O/r03_pytest12b.log 2>&1; echo "parity K=4 rc=$?" >> $O/r03_pytest12.summary
{ for tl in 0 "" 4 8; do echo "## GF2BV_TWO_LEVEL=$tl"; GF2BV_TWO_LEVEL=$tl timeout 120 python tools/profile_one.py 131072 3 | tail -2; GF2BV_TWO_LEVEL=$tl timeout 200 python tools/profile_one.py 262144 3 | tail -2; done
  for w in 224 248 256; do echo "## GF2BV_TWO_LEVEL=8 GF2BV_OUTER_WGS=$w"; GF2BV_OUTER_WGS=$w GF2BV_TWO_LEVEL=8 timeout 120 python tools/profile_one.py 131072 3 | tail -1; GF2BV_OUTER_WGS=$w GF2BV_TWO_LEVEL=8 timeout 200 python tools/profile_one.py 262144 2 | tail -1; done
  echo "## 65536 forced"; for tl in 2 4; do GF2BV_TWO_LEVEL=$tl python tools/profile_one.py 65536 3 | tail -1; done; } > $O/r03_two_level_times4.txt 2>&1
cd /tmp; export TMPDIR=/tmp
GF2BV_TWO_LEVEL=8 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r03_tl8_131072 -- python $R/tools/profile_one.py 131072 1 > $O/r03_tl8_131072.log 2>&1
