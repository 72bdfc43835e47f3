#!/bin/bash
O=gpurun_out; mkdir -p $O
{
echo "# 262144^2 (seed 1242), elimination ms by super-panel size P (outer panels of 12 blocks) and Strassen levels; two-level = GF2BV_THREE_LEVEL=0"
GF2BV_THREE_LEVEL=0 SEED=1242 RESIDUAL=0 timeout 600 python tools/profile_one.py 262144 2 | tail -1
for P in 2 3 4 5 6 8 10 14 20; do
  for L in 0 1 d; do
    if [ $L == d ]; then unset GF2BV_STRASSEN; else export GF2BV_STRASSEN=$L; fi
    echo "## P=$P L=$L"; GF2BV_THREE_LEVEL=$P SEED=1242 RESIDUAL=0 timeout 600 python tools/profile_one.py 262144 2 | tail -1
  done
done
unset GF2BV_STRASSEN
echo "# threshold scan at P=10 and P=4 (MiB right of the super-panel)"
for P in 4 10; do for m in 512 2048 4096; do echo "## P=$P min=$m"; GF2BV_THREE_LEVEL=$P GF2BV_THREE_LEVEL_MIN_MIB=$m SEED=1242 RESIDUAL=0 timeout 600 python tools/profile_one.py 262144 2 | tail -1; done; done
} > $O/r05_sp_scan.txt 2>&1
