#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_small.py -x -q -s > $O/r03_pytest19a.log 2>&1; echo "small rc=$?" > $O/r03_pytest19.summary
timeout 1800 python -m pytest tests -m gpu -x -q > $O/r03_pytest19.log 2>&1; echo "full suite rc=$?" >> $O/r03_pytest19.summary
{ echo "## default plan"; for n in 65536 98304 131072 196608 262144; do python tools/profile_one.py $n 3 | tail -1; done
  echo "## host submission (GF2BV_TRACE)"; for n in 8192 16384 32768 65536; do GF2BV_TRACE=1 python tools/profile_one.py $n 2 2>&1 | grep -E "submitted|enqueue_forward|^N=" | tail -3; done
  echo "## small systems"; python tools/small_latency.py 2>&1 | tail -12; } > $O/r03_times19.txt 2>&1
python bench.py > $O/r03_bench_default2.json 2> $O/r03_bench_default2.err
