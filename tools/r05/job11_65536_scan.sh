#!/bin/bash
# round 5: can outer panels of FEW blocks pay at the 65536^2 headline (round 3 scanned K = 8 only: never)?
O=gpurun_out; mkdir -p $O
{
echo "# 65536^2 seed 1234, tools/profile_one.py 65536 4 (last line = warm), elimination ms; default = one level"
RESIDUAL=0 python tools/profile_one.py 65536 4 | tail -2
for K in 2 3 4 6; do for m in 128 192 256 320 384; do
  echo "## K=$K min_mib=$m"; GF2BV_OUTER_K=$K GF2BV_TWO_LEVEL_MIN_MIB=$m RESIDUAL=0 python tools/profile_one.py 65536 4 | tail -2
done; done
echo "# 131072^2: K = 8 (default below 3 GiB) against 12 and smaller thresholds"
SEED=1242 RESIDUAL=0 python tools/profile_one.py 131072 3 | tail -1
for K in 4 6 12; do for m in 256 512; do echo "## K=$K min_mib=$m"; SEED=1242 GF2BV_OUTER_K=$K GF2BV_TWO_LEVEL_MIN_MIB=$m RESIDUAL=0 python tools/profile_one.py 131072 3 | tail -1; done; done
} > $O/r05_65536_scan.txt 2>&1
