#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_stress.py -x -q -k "sparse_block" > $O/r05_job9_tests.log 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "mt19937 or golden or sparse or density or nlfsr" >> $O/r05_job9_tests.log 2>&1
python tools/mt_stats.py 32 17 9 1 1337 137 > $O/r05_mt_stats_sparse.txt 2>&1
python tools/probe_sparse.py 32 17 137 > $O/r05_sparse_probe.txt 2>&1
