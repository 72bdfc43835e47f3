#!/bin/bash
# round 5, lease 3: the three-level elimination -- parity on small systems (forced), then timing at the sizes it is planned for
O=gpurun_out; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_stress.py -x -q -k "three_level" > $O/r05_job3_tests.log 2>&1
{
for n in 131072 262144; do
  echo "## N=$n two-level (GF2BV_THREE_LEVEL=0)"; GF2BV_THREE_LEVEL=0 SEED=1242 timeout 600 python tools/profile_one.py $n 2
  echo "## N=$n three-level default"; SEED=1242 timeout 600 python tools/profile_one.py $n 2
  echo "## N=$n three-level, TIME_KERNELS"; TIME_KERNELS=1 SEED=1242 timeout 600 python tools/profile_one.py $n 2
  for L in 0 1 2 3; do echo "## N=$n three-level GF2BV_STRASSEN=$L"; GF2BV_STRASSEN=$L SEED=1242 timeout 600 python tools/profile_one.py $n 2; done
done
} > $O/r05_three_level_first.txt 2>&1
timeout 1200 python -m pytest tests/test_gpu_stress.py tests/test_gpu_parity.py -x -q > $O/r05_job3_tests_all.log 2>&1
