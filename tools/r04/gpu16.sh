#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd $R
s=$(date +%s.%N); python bench.py > $O/r04_bench_b.json 2> $O/r04_bench_b.err; echo "bench rc=$? seconds $(echo "$(date +%s.%N) - $s" | bc)" > $O/r04_gpu16.summary
