#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd $R
{ for u in 0 6 7 0 6; do echo "## GF2BV_UPDATE=$u"; GF2BV_UPDATE=$u TIME_KERNELS=1 timeout 120 python tools/profile_one.py 65536 4 | tail -2; GF2BV_UPDATE=$u timeout 120 python tools/profile_one.py 65536 3 | tail -1; GF2BV_UPDATE=$u timeout 120 python tools/profile_one.py 131072 3 | tail -1; done; } > $O/r03_depth59.txt 2>&1
