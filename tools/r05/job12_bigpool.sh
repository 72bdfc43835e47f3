#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_stress.py -x -q -k "sparse_block" > $O/r05_job12_tests.log 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "mt19937" >> $O/r05_job12_tests.log 2>&1
python tools/mt_stats.py 1 9 32 > $O/r05_mt_bigpool.txt 2>&1
