#!/bin/bash
# round 5, lease 1: pool hygiene test + a parity subset on the rebuilt library; batch K-fusion A/B on the FINAL gang layout (XCD pinning,
# streaming accesses, gangs of 32); kernel stats of the C3 (MT19937) and C5 (xoshiro) legs.
O=gpurun_out; mkdir -p $O
{
timeout 900 python -m pytest tests/test_gpu_batch_c4.py -x -q -k "pool or share" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "gang or update_configurations or golden" 2>&1 | tail -5
} > $O/r05_job1_tests.log 2>&1
{
echo "# batch K-fusion A/B, final round-4 layout (gangs of 32, a system per XCD, streaming row accesses)"
for n in 192 512; do
  echo "## $n x 32768^2, one-level (default)"; timeout 600 python tools/batch_time.py 32768 $n 3
  echo "## $n x 32768^2, GF2BV_GANG_TWO_LEVEL=1 (K = 8)"; GF2BV_GANG_TWO_LEVEL=1 timeout 600 python tools/batch_time.py 32768 $n 3
done
echo "## 192, two-level K=4 / K=12 / threshold 128 MiB"
GF2BV_GANG_TWO_LEVEL=1 GF2BV_OUTER_K=4 timeout 600 python tools/batch_time.py 32768 192 3
GF2BV_GANG_TWO_LEVEL=1 GF2BV_OUTER_K=12 timeout 600 python tools/batch_time.py 32768 192 3
GF2BV_GANG_TWO_LEVEL=1 GF2BV_TWO_LEVEL_MIN_MIB=128 timeout 600 python tools/batch_time.py 32768 192 3
echo "## 192, gangs of 8 (ONE system per XCD at a time) one-level / two-level"
GF2BV_GANG=8 timeout 600 python tools/batch_time.py 32768 192 3
GF2BV_GANG=8 GF2BV_GANG_TWO_LEVEL=1 timeout 600 python tools/batch_time.py 32768 192 3
} > $O/r05_batch_scans.txt 2>&1
JOB_TIMEOUT=600 bash tools/jobs/kernel_stats.sh r05_c3_mt32 python tools/mt_stats.py 32
JOB_TIMEOUT=600 bash tools/jobs/kernel_stats.sh r05_c3_mt1 python tools/mt_stats.py 1
JOB_TIMEOUT=600 bash tools/jobs/kernel_stats.sh r05_c3_mt17 python tools/mt_stats.py 17
JOB_TIMEOUT=600 bash tools/jobs/kernel_stats.sh r05_c5_xoshiro python examples/xoshiro_recovery.py
python tools/mt_stats.py 32 17 9 1 1337 137 > $O/r05_mt_stats_before.txt 2>&1
