#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
{ echo "# K-loop costing, second run: every variant on the same matrix contents, two rounds ($(date -u))"; $R/tools/_probe/mbk 98304 512
  echo "# product kernel, same box and size"; MB_ONLY=512,3,1 $R/tools/_probe/mbu16r3_full 98304 512 256; MB_ONLY=512,3,1 $R/tools/_probe/mbu16_l2 98304 512 256
  echo "# 196608 rows x 256 tiles"; MB_NOCHECK=1 $R/tools/_probe/mbk 196608 256 | grep "S=16"; MB_ONLY=512,3,1 $R/tools/_probe/mbu16r3_full 196608 256 256; } > $O/r03_kloop2.txt 2>&1
cd $R
{ for lib in "" $R/tools/_probe/libgf2bv_hip_ntl.so; do echo "## GF2BV_LIB=$lib"; for n in 32768 65536 131072; do GF2BV_LIB=$lib python tools/profile_one.py $n 4 | tail -2; done; GF2BV_LIB=$lib python tools/profile_one.py 262144 2 | tail -1; done; } > $O/r03_ntl_insolve.txt 2>&1
