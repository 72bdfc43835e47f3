#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd $R
{ for m in 512 768 1024 1536; do echo "## K=12 GF2BV_TWO_LEVEL_MIN_MIB=$m"; for n in 196608 262144; do GF2BV_OUTER_K=12 GF2BV_TWO_LEVEL_MIN_MIB=$m timeout 200 python tools/profile_one.py $n 3 | tail -1; done; done
  echo "## K=12 393216 / 524288"; GF2BV_OUTER_K=12 timeout 300 python tools/largest_run.py 393216 | tail -1; GF2BV_OUTER_K=8 timeout 300 python tools/largest_run.py 393216 | tail -1; } > $O/r03_k12_63.txt 2>&1
