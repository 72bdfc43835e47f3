#!/bin/bash
# Round 6: k_outer_apply (P = T x S) on sixteen wavefronts x 3 pivots per lane instead of eight x 6 -- parity under forced two-level plans, then A/B against a -DGF2_OA_NT=512 build
mkdir -p gpurun_out /tmp/dbg
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DGF2_OA_NT=512 gf2bv_amd/csrc/gf2_solver.hip -o /tmp/dbg/liboa512.so
( timeout 900 python -m pytest tests/test_gpu_stress.py tests/test_gpu_parity.py -x -q -m gpu -k "two_level or outer or target_262144 or pivotless or large" 2>&1 | tail -3 ) > gpurun_out/r06k_pytest.log 2>&1
cat gpurun_out/r06k_pytest.log
run() { echo "## $*"; env "$@" SEED=1242 python tools/profile_one.py 262144 3 | tail -2 | cut -c1-150; env "$@" python tools/profile_one.py 131072 3 | tail -2 | cut -c1-150; }
( run GF2BV_X=1024; run GF2BV_LIB=/tmp/dbg/liboa512.so; run GF2BV_X=1024; run GF2BV_LIB=/tmp/dbg/liboa512.so ) > gpurun_out/r06k_apply_ab.txt 2>&1
cat gpurun_out/r06k_apply_ab.txt
