#!/bin/bash
O=gpurun_out; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DGF2_SPARSE_DEBUG gf2bv_amd/csrc/gf2_solver.hip -o /tmp/libdbg.so
GF2BV_LIB=/tmp/libdbg.so python tools/mt_stats.py 32 17 9 1337 137 > $O/r05_mt_rounds.txt 2>&1
