#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd $R
{ for lib in gf2bv_amd/libgf2bv_hip.so tools/_probe/lib_dgf2_nt_loaddgf2_nt_store.so gf2bv_amd/libgf2bv_hip.so tools/_probe/lib_dgf2_nt_loaddgf2_nt_store.so; do echo "## in-solve, bulk launches bracketed by events: $lib"
    for n in 32768 65536 98304; do TIME_KERNELS=1 GF2BV_LIB=$R/$lib timeout 120 python tools/profile_one.py $n 4 | tail -2; done; done; } > $O/r03_nt41.txt 2>&1
