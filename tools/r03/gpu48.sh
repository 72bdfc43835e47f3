#!/bin/bash
# look-ahead back behind a k_gate launch; retry counter in the stats; large-size tests
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -x -q --durations=8 > $O/r03_pytest48.log 2>&1; echo "full suite rc=$?" > $O/r03_final48.summary
{ for n in 8192 32768 65536 131072 262144; do timeout 200 python tools/profile_one.py $n 3 | tail -1; done
  GF2BV_TRACE=1 timeout 600 python tools/largest_run.py 393216 2>&1 | grep -E "enqueue_forward|^N="
  GF2BV_TRACE=1 timeout 600 python tools/largest_run.py 2>&1 | grep -E "enqueue_forward|^N="; } > $O/r03_times48.txt 2>&1
timeout 400 python tests/manual/soak_concurrent.py 6 16 5 > $O/r03_soak48.log 2>&1; echo "soak rc=$?" >> $O/r03_final48.summary
