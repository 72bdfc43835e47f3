#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd $R
{ for pad in 2 8 64 2 8 64; do echo "## GF2BV_SLAB_PAD=$pad"
    for n in 32768 65536; do GF2BV_SLAB_PAD=$pad timeout 120 python tools/profile_one.py $n 4 | tail -2; done
    GF2BV_SLAB_PAD=$pad timeout 120 python tools/profile_one.py 131072 3 | tail -1
  done
  for pad in 2 8 64; do echo "## GF2BV_SLAB_PAD=$pad"; GF2BV_SLAB_PAD=$pad timeout 120 python tools/profile_one.py 262144 3 | tail -2; done; } > $O/r03_slabpad36.txt 2>&1
