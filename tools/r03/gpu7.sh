#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd $R
{ for tl in 0 "" 2 4 8; do echo "## GF2BV_TWO_LEVEL=$tl"; for n in 65536 131072; do GF2BV_TWO_LEVEL=$tl timeout 120 python tools/profile_one.py $n 3 | tail -2; done; GF2BV_TWO_LEVEL=$tl timeout 200 python tools/profile_one.py 262144 3 | tail -2; done; } > $O/r03_two_level_times2.txt 2>&1
GF2BV_TWO_LEVEL=4 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stress.py -x -q > $O/r03_pytest7_k4.log 2>&1; echo "K=4 rc=$?" > $O/r03_pytest7.summary
timeout 900 python -m pytest tests/test_gpu_stress.py tests/test_gpu_slab.py -x -q > $O/r03_pytest7_def.log 2>&1; echo "default rc=$?" >> $O/r03_pytest7.summary
python bench.py --workload sharded --n 65536 --steps 3 --warmup 1 > $O/r03_sharded2_65536.json 2> $O/r03_sharded2_65536.err
cd /tmp; export TMPDIR=/tmp
GF2BV_TWO_LEVEL=4 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r03_tl4_131072 -- python $R/tools/profile_one.py 131072 1 > $O/r03_tl4_131072.log 2>&1
