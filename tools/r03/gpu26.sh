#!/bin/bash
# does the inner path run better beside the outer pass when its kernels FIT beside it?  base = HEAD (k_update16k with 140 KiB of LDS),
# tree = 132 KiB (k_block_fast / k_block_trsm fit), cap64 = tree + k_narrow_all / k_prio_window capped at 64 VGPRs
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd $R
{ for lib in tools/_probe/libgf2bv_base.so gf2bv_amd/libgf2bv_hip.so tools/_probe/libgf2bv_cap64.so; do
    echo "## $lib"
    for n in 32768 65536; do GF2BV_LIB=$R/$lib python tools/profile_one.py $n 4 | tail -2; done
    for n in 131072 262144; do GF2BV_LIB=$R/$lib python tools/profile_one.py $n 3 | tail -2; done
    echo "# 65536 forced K=8"; GF2BV_TWO_LEVEL=8 GF2BV_LIB=$R/$lib python tools/profile_one.py 65536 3 | tail -2
    echo "# 65536 min_mib=256"; GF2BV_TWO_LEVEL_MIN_MIB=256 GF2BV_LIB=$R/$lib python tools/profile_one.py 65536 3 | tail -2
  done; } > $O/r03_fit26.txt 2>&1
