#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "pinned or panel_kernels or mt19937" > $O/r05_job14.log 2>&1
GF2BV_THREE_LEVEL=10 SEED=1242 timeout 300 python tools/profile_one.py 262144 1 >> $O/r05_job14.log 2>&1
