#!/bin/bash
# diagnostic: how the echelon rounds of k_block_sparse converge on MT19937 with one bit / nine bits per output (survivors and covered columns per round)
# (the -DGF2_SPARSE_ROUNDS block this needs was taken out of gf2_kernels.hip.h again so that the shipped source is the one the closing records were taken on: it is in commit 3df13db)
mkdir -p gpurun_out /tmp/dbg
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DGF2_SPARSE_ROUNDS gf2bv_amd/csrc/gf2_solver.hip -o /tmp/dbg/librounds.so
GF2BV_LIB=/tmp/dbg/librounds.so python tools/mt_stats.py 1 2>&1 | grep "ROUNDS" | sort -u | head -150 > gpurun_out/r06i_rounds_bs1.txt
GF2BV_LIB=/tmp/dbg/librounds.so python tools/mt_stats.py 9 2>&1 | grep "ROUNDS" | sort -u | head -100 > gpurun_out/r06i_rounds_bs9.txt
wc -l gpurun_out/r06i_rounds_bs*.txt; head -40 gpurun_out/r06i_rounds_bs1.txt
