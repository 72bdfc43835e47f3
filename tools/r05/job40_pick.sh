#!/bin/bash
# does the low-priority stream a single solve uses matter for the dense headline too?  k-th created stream, one process each
cd /root/repo; mkdir -p gpurun_out
{
for k in 0 1 2 3 4 5 6 7; do
  echo "== GF2BV_LOW_PICK=$k"
  GF2BV_LOW_PICK=$k GF2BV_TRACE=1 timeout 300 python tools/profile_one.py 65536 4 2>&1 | grep "stream pair\|N=" | cut -c1-110 | tail -n 5
  GF2BV_LOW_PICK=$k timeout 300 python tools/mt_stats.py 32 2>&1 | tail -n 1 | cut -c1-120
done
} > gpurun_out/r05_pick.txt
