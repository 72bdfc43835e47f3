#!/bin/bash
# search + narrow step in one launch (k_block_fast_narrow): parity, then times against GF2BV_FUSED_NARROW=0
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q > $O/r03_pytest29a.log 2>&1; echo "parity rc=$?" > $O/r03_pytest29.summary
{ for f in 1 0; do echo "## GF2BV_FUSED_NARROW=$f"
    for n in 8192 16384 32768 65536; do GF2BV_FUSED_NARROW=$f timeout 120 python tools/profile_one.py $n 4 | tail -2; done
    GF2BV_FUSED_NARROW=$f timeout 120 python tools/profile_one.py 131072 3 | tail -1
  done; } > $O/r03_fused29.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q > $O/r03_pytest29b.log 2>&1; echo "full suite rc=$?" >> $O/r03_pytest29.summary
