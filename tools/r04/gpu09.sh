#!/bin/bash
# round 4: k_small_solve: parity on both paths, then the whole GPU suite, then latencies
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_small.py -x -q -s > $O/r04_pytest09a.log 2>&1; echo "small rc=$?" > $O/r04_gpu09.summary
timeout 2400 python -m pytest tests -m gpu -q > $O/r04_pytest09b.log 2>&1; echo "full suite rc=$?" >> $O/r04_gpu09.summary
{ for s in 1 0; do echo "## GF2BV_SMALL=$s"; GF2BV_SMALL=$s timeout 120 python tools/small_latency.py; done; } > $O/r04_small_latency.txt 2>&1
