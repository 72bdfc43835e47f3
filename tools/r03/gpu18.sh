#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd $R
{ for n in 65536 98304 131072; do for mib in 100000 128 192 256 384 512 768 1024; do echo "## N=$n min_mib=$mib"; GF2BV_TWO_LEVEL_MIN_MIB=$mib python tools/profile_one.py $n 4 | tail -2; done; done
  echo "## host submission vs device (GF2BV_TRACE)"; for n in 16384 32768 65536; do GF2BV_TRACE=1 python tools/profile_one.py $n 3 2>&1 | tail -8; done; } > $O/r03_threshold_scan.txt 2>&1
