#!/bin/bash
# k_prio_window as its own gate (no k_gate launch before it): parity, then times against GF2BV_PRIO_GATE=1
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd $R
rm -f $O/r03_pytest32.summary
for i in 1 2; do timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "handover or fallbacks or optimistic" > $O/r03_pytest32a_$i.log 2>&1; echo "handover x$i rc=$?" >> $O/r03_pytest32.summary; done
{ for f in 0 1 0 1; do echo "## GF2BV_PRIO_GATE=$f"
    for n in 8192 32768 65536; do GF2BV_PRIO_GATE=$f timeout 120 python tools/profile_one.py $n 4 | tail -2; done
  done; GF2BV_PRIO_GATE=0 python tools/profile_one.py 131072 3 | tail -1;  GF2BV_PRIO_GATE=1 python tools/profile_one.py 131072 3 | tail -1; } > $O/r03_priogate32.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q > $O/r03_pytest32b.log 2>&1; echo "full suite rc=$?" >> $O/r03_pytest32.summary
timeout 600 python tests/manual/stress_parity.py 150 43 > $O/r03_stress32.log 2>&1; echo "stress rc=$?" >> $O/r03_pytest32.summary
