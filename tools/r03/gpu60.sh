#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stress.py -x -q > $O/r03_pytest60.log 2>&1; echo "parity+stress rc=$?" > $O/r03_final60.summary
timeout 300 python tests/manual/soak_concurrent.py 6 16 13 > $O/r03_soak60.log 2>&1; echo "soak rc=$?" >> $O/r03_final60.summary
{ for lib in tools/_probe/lib_head.so gf2bv_amd/libgf2bv_hip.so tools/_probe/lib_head.so gf2bv_amd/libgf2bv_hip.so; do echo "## $lib"
    for n in 8192 16384 32768 65536; do GF2BV_LIB=$R/$lib timeout 120 python tools/profile_one.py $n 5 | tail -2; done; done; } > $O/r03_flag60.txt 2>&1
