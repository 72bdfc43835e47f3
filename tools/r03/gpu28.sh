#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd $R; timeout 600 python tools/_probe/target_vs_bench.py > $O/r03_target_vs_bench.txt 2>&1
