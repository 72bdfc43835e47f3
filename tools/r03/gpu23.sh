#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd $R
{ for e in 0 1 2 3; do echo "## experiment $e"; for K in 4 8; do ./tools/_mbk_e$e 131072 1024 $K; done; done; } > $O/r03_mbk23.txt 2>&1
