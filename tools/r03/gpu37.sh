#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd $R
{ for pad in 2 64 128 256 320 1024 2 64; do echo "## GF2BV_SLAB_PAD=$pad"
    for n in 65536 49152 98304; do GF2BV_SLAB_PAD=$pad timeout 120 python tools/profile_one.py $n 4 | tail -2; done
  done; } > $O/r03_slabpad37.txt 2>&1
