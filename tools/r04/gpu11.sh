#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd $R
{ for s in 1 0; do echo "## GF2BV_SMALL=$s"; GF2BV_SMALL=$s timeout 300 python tools/small_latency.py 300x200 400x384 520x512 700x640 800x767 1200x511 2000x300 4096x191 4096x100 3000x129; done; } > $O/r04_small_mid.txt 2>&1
