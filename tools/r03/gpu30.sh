#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd $R
for i in 1 2 3; do timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "handover or fallbacks or optimistic" > $O/r03_pytest30a_$i.log 2>&1; echo "handover x$i rc=$?" >> $O/r03_pytest30.summary; done
timeout 1500 python -m pytest tests -m gpu -x -q > $O/r03_pytest30b.log 2>&1; echo "full suite rc=$?" >> $O/r03_pytest30.summary
timeout 900 python tests/manual/stress_parity.py 200 41 > $O/r03_stress30.log 2>&1; echo "stress rc=$?" >> $O/r03_pytest30.summary
{ for f in 1 0; do echo "## GF2BV_FUSED_NARROW=$f"
    for n in 8192 32768 65536; do GF2BV_FUSED_NARROW=$f timeout 120 python tools/profile_one.py $n 4 | tail -2; done
  done; } > $O/r03_fused30.txt 2>&1
