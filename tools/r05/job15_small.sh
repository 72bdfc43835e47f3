#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_small.py -x -q > $O/r05_job15.log 2>&1
python tools/small_latency.py 4x4 128x128 640x256 800x767 4096x191 >> $O/r05_job15.log 2>&1
python examples/xoshiro_recovery.py >> $O/r05_job15.log 2>&1
