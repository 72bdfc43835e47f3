#!/bin/bash
O=gpurun_out; mkdir -p $O
{
for w in 0 32 64 128 192; do echo "## GF2BV_INNER_WGS=$w"; GF2BV_INNER_WGS=$w SEED=1242 RESIDUAL=0 python tools/profile_one.py 262144 3 | tail -n 2; done
for w in 0 64 128; do echo "## 131072 GF2BV_INNER_WGS=$w"; GF2BV_INNER_WGS=$w SEED=1242 RESIDUAL=0 python tools/profile_one.py 131072 3 | tail -n 1; done
} > $O/r05_inner_wgs.txt 2>&1
