#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "k_update16k" --kernel-iteration-range "[1-6]" --kernel-trace --output-format csv -d $O/r03_fetch_k16k_262144 -- python $R/tools/profile_one.py 262144 1 > $O/r03_fetch_k16k_262144.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "k_update16k" --kernel-iteration-range "[1-6]" --kernel-trace --output-format csv -d $O/r03_write_k16k_262144 -- python $R/tools/profile_one.py 262144 1 > $O/r03_write_k16k_262144.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-include-regex "k_update16k" --kernel-iteration-range "[1-6]" --kernel-trace --output-format csv -d $O/r03_sq_k16k_262144 -- python $R/tools/profile_one.py 262144 1 > $O/r03_sq_k16k_262144.log 2>&1
cd $R; timeout 1200 python tests/manual/stress_parity.py 600 77 > $O/r03_stress21.log 2>&1; echo "stress rc=$?" > $O/r03_pytest21.summary
