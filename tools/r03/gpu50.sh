#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -m gpu -x -q > $O/r03_pytest50.log 2>&1; echo "full suite rc=$?" > $O/r03_final50.summary
timeout 400 python tests/manual/soak_concurrent.py 6 16 9 > $O/r03_soak50.log 2>&1; echo "soak rc=$?" >> $O/r03_final50.summary
timeout 400 python tests/manual/stress_parity.py 180 59 > $O/r03_stress50.log 2>&1; echo "stress rc=$?" >> $O/r03_final50.summary
{ for n in 8192 32768 65536 65536 131072 262144; do timeout 200 python tools/profile_one.py $n 4 | tail -1; done
  echo "## GF2BV_FUSED_RPT=1"; for n in 65536; do GF2BV_FUSED_RPT=1 timeout 200 python tools/profile_one.py $n 4 | tail -2; done
  echo "## GF2BV_FUSED_NARROW=0"; for n in 32768 65536; do GF2BV_FUSED_NARROW=0 timeout 200 python tools/profile_one.py $n 4 | tail -2; done; } > $O/r03_times50.txt 2>&1
