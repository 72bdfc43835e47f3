#!/bin/bash
# compact pivot rows from k_block_trsm (Pc): parity, then kernel time and wall against GF2BV_PC=0
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stress.py tests/test_gpu_slab.py -x -q > $O/r03_pytest54a.log 2>&1; echo "parity+stress+slab rc=$?" > $O/r03_final54.summary
{ for f in 1 0 1 0; do echo "## GF2BV_PC=$f"; for n in 32768 65536; do GF2BV_PC=$f TIME_KERNELS=1 timeout 120 python tools/profile_one.py $n 4 | tail -2; done; done
  for f in 1 0; do echo "## GF2BV_PC=$f, no event brackets"; for n in 16384 65536 131072; do GF2BV_PC=$f timeout 120 python tools/profile_one.py $n 4 | tail -2; done; done; } > $O/r03_pc54.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q > $O/r03_pytest54b.log 2>&1; echo "full suite rc=$?" >> $O/r03_final54.summary
