#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout 1300 python tests/manual/stress_parity.py 1200 5150 > $O/r05_stress_long.log 2>&1; echo "rc=$?" >> $O/r05_stress_long.log
timeout 400 python tests/manual/soak_concurrent.py 6 8 51 > $O/r05_soak.log 2>&1; echo "rc=$?" >> $O/r05_soak.log
