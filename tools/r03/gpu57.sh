#!/bin/bash
# k_update16: the span's first batches requested before its tables are built -- parity, then same-box A/B against the previous build
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stress.py -x -q > $O/r03_pytest57.log 2>&1; echo "parity+stress rc=$?" > $O/r03_final57.summary
{ for lib in tools/_probe/lib_head.so gf2bv_amd/libgf2bv_hip.so tools/_probe/lib_head.so gf2bv_amd/libgf2bv_hip.so; do echo "## $lib"
    for n in 32768 65536; do GF2BV_LIB=$R/$lib TIME_KERNELS=1 timeout 120 python tools/profile_one.py $n 4 | tail -2; done
    GF2BV_LIB=$R/$lib timeout 120 python tools/profile_one.py 65536 4 | tail -2
    GF2BV_LIB=$R/$lib timeout 120 python tools/profile_one.py 131072 3 | tail -1; done; } > $O/r03_early57.txt 2>&1
