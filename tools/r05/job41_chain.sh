#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
{
for k in 0 1 2 3 4 5 6 7 8; do
  echo "== GF2BV_LOW_PICK=$k"
  GF2BV_CHAIN_PROBE=1 GF2BV_LOW_PICK=$k GF2BV_TRACE=1 timeout 300 python tools/mt_stats.py 32 2>&1 | grep "chain probe\|stream pair\|bs=32" | cut -c1-150
done
} > gpurun_out/r05_chain.txt
