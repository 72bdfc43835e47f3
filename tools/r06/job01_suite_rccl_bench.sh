#!/bin/bash
# Round 6, first lease: the RCCL init A/B, the whole GPU suite in the driver's own form, the default bench line.
mkdir -p gpurun_out
python tools/r06/rccl_init_ab.py "lease $(date +%H%M)" >> gpurun_out/r06_rccl_init.txt 2>&1
( time python -m pytest tests/ -x -q -m gpu --durations=15 ) > gpurun_out/r06a_pytest.log 2>&1
tail -30 gpurun_out/r06a_pytest.log
python bench.py > gpurun_out/r06a_bench.json 2> gpurun_out/r06a_bench.err
tail -c 600 gpurun_out/r06a_bench.json
cat gpurun_out/r06_rccl_init.txt
