#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_batch_c4.py tests/test_gpu_parity.py -q -x -k "c4 or gang or batch or pool" > $O/r05_job22.log 2>&1
timeout 600 python tools/batch_time.py 32768 64 4 >> $O/r05_job22.log 2>&1
timeout 600 python tools/batch_time.py 32768 512 3 >> $O/r05_job22.log 2>&1
timeout 600 python tools/batch_time.py 4096 64 4 >> $O/r05_job22.log 2>&1
timeout 600 python tools/batch_time.py 8192 12 4 >> $O/r05_job22.log 2>&1
