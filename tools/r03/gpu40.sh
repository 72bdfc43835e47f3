#!/bin/bash
# cache-policy bits again, now that every batch is whole lines (slabs padded to KiB)
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd $R
{ for v in plain dgf2_nt_store dgf2_nt_load dgf2_nt_loaddgf2_nt_store plain; do echo "## isolated $v"; for a in "65536 512" "131072 512" "65536 256"; do MB_ONLY=512,3,1 ./tools/_mb16_$v $a 256 | grep "NT="; done; done
  for lib in gf2bv_amd/libgf2bv_hip.so tools/_probe/lib_dgf2_nt_store.so tools/_probe/lib_dgf2_nt_loaddgf2_nt_store.so gf2bv_amd/libgf2bv_hip.so; do echo "## in-solve $lib"
    for n in 65536 131072; do GF2BV_LIB=$R/$lib timeout 120 python tools/profile_one.py $n 4 | tail -2; done; done; } > $O/r03_nt40.txt 2>&1
