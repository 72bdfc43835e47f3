#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_stress.py -q -x -k "three_level" > $O/r05_job19.log 2>&1
GF2BV_TWO_LEVEL=2 GF2BV_THREE_LEVEL=2 timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x >> $O/r05_job19.log 2>&1
GF2BV_THREE_LEVEL=10 SEED=1242 timeout 300 python tools/profile_one.py 262144 2 >> $O/r05_job19.log 2>&1
