#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd $R
{ for t in 4 8 16 32 64 128 256 512; do echo "## 65536 rows x $t tiles"; MB_ONLY=512,3,1 ./tools/_mb16 65536 $t 256 | grep "NT="; done
  for t in 8 32 128; do echo "## 32768 rows x $t tiles"; MB_ONLY=512,3,1 ./tools/_mb16 32768 $t 256 | grep "NT="; done; } > $O/r03_fixed42.txt 2>&1
