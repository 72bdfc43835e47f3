#!/bin/bash
# phases inside the one-launch block search (k_block_fast_narrow's workgroup 0) at three sizes
mkdir -p gpurun_out tools/_probe
( python tools/probe_fast.py 65536 8 128 200 240; python tools/probe_fast.py 32768 8 64 120; python tools/probe_fast.py 16384 8 40 ) 2>&1 | tail -12 > gpurun_out/r06g_probe_fast.txt
cat gpurun_out/r06g_probe_fast.txt
