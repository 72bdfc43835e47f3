#!/bin/bash
# Round 6: k_block_sparse without the per-panel compaction while the first-64 attempt is off -- parity of the sparse paths, then the six MT19937 systems
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_stress.py tests/test_gpu_parity.py tests/test_gpu_packed.py -x -q -m gpu -k "sparse or mt19937 or nlfsr or golden or randomised" 2>&1 | tail -4 ) > gpurun_out/r06h_pytest.log 2>&1
cat gpurun_out/r06h_pytest.log
python tools/mt_stats.py 32 17 9 1 1337 137 2>&1 | grep -v "gives up" > gpurun_out/r06h_mt_stats.txt
cat gpurun_out/r06h_mt_stats.txt
timeout 400 python tests/manual/stress_parity.py 300 909 2>&1 | tail -2
