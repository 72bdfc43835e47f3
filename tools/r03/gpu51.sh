#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd $R
{ echo "## default (two streams)"; TIME_KERNELS=1 python tools/profile_one.py 65536 4 | tail -2
  echo "## GF2BV_SERIAL=1 (one stream: the bulk update runs alone)"; GF2BV_SERIAL=1 TIME_KERNELS=1 python tools/profile_one.py 65536 4 | tail -2
  echo "## GF2BV_FUSED_NARROW=0"; GF2BV_FUSED_NARROW=0 TIME_KERNELS=1 python tools/profile_one.py 65536 4 | tail -2
  echo "## 131072 default / serial"; TIME_KERNELS=1 python tools/profile_one.py 131072 3 | tail -1; GF2BV_SERIAL=1 TIME_KERNELS=1 python tools/profile_one.py 131072 3 | tail -1; } > $O/r03_serial51.txt 2>&1
