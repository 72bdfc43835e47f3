#!/bin/bash
# round 5, lease 2: Strassen-Winograd over the M4RM base case (tools/microbench_strassen.hip) + the pool test again
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_batch_c4.py -x -q -k "pool" > $O/r05_job2_pool.log 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench_strassen.hip -o /tmp/mbs > $O/r05_strassen_build.log 2>&1
{
echo "## small, with the host check of the base case"
MB_HOSTCHECK=1 timeout 300 /tmp/mbs 16384 8 8 2
MB_HOSTCHECK=1 timeout 300 /tmp/mbs 32768 16 16 2
echo "## C(N x N/2) ^= A(N x N/2) . B(N/2 x N/2), N = 262144 (the product VERDICT round 4 names)"
timeout 900 /tmp/mbs 262144 1024 512 3
echo "## Schur update of a super-panel of 32768 / 16384 / 8192 columns on a 196608 x 196608 trailing matrix"
timeout 900 /tmp/mbs 196608 1536 128 3
timeout 900 /tmp/mbs 196608 1536 64 3
timeout 900 /tmp/mbs 196608 1536 32 2
echo "## smaller trailing matrices"
timeout 900 /tmp/mbs 131072 1024 128 3
timeout 900 /tmp/mbs 65536 512 128 3
timeout 900 /tmp/mbs 65536 512 64 2
} > $O/r05_strassen.txt 2>&1
