#!/bin/bash
# experiment: the selection rounds of k_block_sparse with the HIGHEST set bit as a candidate's pivot instead of the lowest (-DGF2_SP_HIGH)
mkdir -p gpurun_out /tmp/dbg
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DGF2_SP_HIGH gf2bv_amd/csrc/gf2_solver.hip -o /tmp/dbg/libhigh.so
( echo "## lowest bit (shipped)"; python tools/mt_stats.py 17 9 1 137 2>&1 | grep -v "gives up"; echo "## highest bit"; GF2BV_LIB=/tmp/dbg/libhigh.so python tools/mt_stats.py 17 9 1 137 2>&1 | grep -v "gives up" ) > gpurun_out/r06j_high.txt
cat gpurun_out/r06j_high.txt
