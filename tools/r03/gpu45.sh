#!/bin/bash
# final build: largest system (524288^2, 64 GiB resident), long randomised differential run, concurrency soak
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 600 python tools/largest_run.py > $O/r03_largest.txt 2>&1; echo "largest rc=$?" > $O/r03_final45.summary
GF2BV_TWO_LEVEL=0 timeout 600 python tools/largest_run.py 393216 >> $O/r03_largest.txt 2>&1
timeout 600 python tools/largest_run.py 393216 >> $O/r03_largest.txt 2>&1
timeout 800 python tests/manual/stress_parity.py 600 53 > $O/r03_stress45.log 2>&1; echo "stress rc=$?" >> $O/r03_final45.summary
timeout 600 python tests/manual/soak_concurrent.py 10 24 11 > $O/r03_soak45.log 2>&1; echo "soak rc=$?" >> $O/r03_final45.summary
