#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd $R
{ for v in a b a b; do echo "## variant $v (a = 32 B per row interleaved, b = two planes; b's results are wrong by construction: timing only)"; MB_ONLY=512,3,1 ./tools/_mb16$v 65536 512 256 | grep "NT="; MB_ONLY=512,3,1 ./tools/_mb16$v 131072 512 256 | grep "NT="; MB_ONLY=512,3,1 ./tools/_mb16$v 65536 256 256 | grep "NT="; done; } > $O/r03_planes39.txt 2>&1
