#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd $R
{ for pad in 2 0 18 66 130 258 274 1026 4098 4114; do echo "## MB_PAD=$pad"; MB_PAD=$pad MB_ONLY=512,3,1 ./tools/_mb16 65536 512 256 | grep "NT="; MB_PAD=$pad MB_ONLY=512,3,1 ./tools/_mb16 131072 512 256 | grep "NT="; done; } > $O/r03_pad35.txt 2>&1
