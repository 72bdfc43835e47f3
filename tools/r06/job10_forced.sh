#!/bin/bash
# Round 6: the parity file (both heuristics settings) with plans FORCED onto every test -- two-level K = 2 / 3 / 8, sparse search off, events -- on the final build;
# then the whole suite once more on this (fresh) box.
mkdir -p gpurun_out; O=gpurun_out/r06f_forced.txt; : > $O
for env in "GF2BV_TWO_LEVEL=2" "GF2BV_TWO_LEVEL=3" "GF2BV_TWO_LEVEL=8" "GF2BV_SPARSE_FAST=0" "GF2BV_FLAG_SYNC=0" "GF2BV_TWO_LEVEL=4 GF2BV_OUTER_SIDE=0"; do
  echo "## $env" >> $O
  ( env $env timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider 2>&1 | tail -2 ) >> $O 2>&1
done
cat $O
( time timeout 1500 python -m pytest tests/ -x -q -m gpu ) 2>&1 | tail -6 > gpurun_out/r06f_suite_again.txt
cat gpurun_out/r06f_suite_again.txt
