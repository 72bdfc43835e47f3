#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_stress.py -x -q -k "two_level" > $O/r03_pytest67.log 2>&1; echo "two-level stress rc=$?" > $O/r03_final67.summary
GF2BV_TWO_LEVEL=12 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "not 262144 and not 327680" > $O/r03_pytest67b.log 2>&1; echo "parity K=12 rc=$?" >> $O/r03_final67.summary
GF2BV_TWO_LEVEL=3 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "not 262144 and not 327680" > $O/r03_pytest67c.log 2>&1; echo "parity K=3 rc=$?" >> $O/r03_final67.summary
{ for lib in tools/_probe/lib_k12.so gf2bv_amd/libgf2bv_hip.so tools/_probe/lib_k12.so gf2bv_amd/libgf2bv_hip.so; do echo "## $lib"; for n in 131072 196608 262144; do GF2BV_LIB=$R/$lib timeout 200 python tools/profile_one.py $n 3 | tail -1; done; done; } > $O/r03_tri67.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "262144 or 327680" > $O/r03_pytest67d.log 2>&1; echo "large properties rc=$?" >> $O/r03_final67.summary
