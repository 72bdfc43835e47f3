#!/bin/bash
O=gpurun_out; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench_strassen.hip -o /tmp/mbs > $O/r05_job16.log 2>&1
MB_HOSTCHECK=1 timeout 300 /tmp/mbs 32768 16 16 2 >> $O/r05_job16.log 2>&1
timeout 300 /tmp/mbs 196608 1536 128 2 >> $O/r05_job16.log 2>&1
timeout 900 python tools/largest_run.py >> $O/r05_job16.log 2>&1
timeout 700 python tests/manual/stress_parity.py 600 777 >> $O/r05_job16.log 2>&1
