#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
{ echo "# r03 stream ceiling, second run: range form with cache policies ($(date -u))"; $R/tools/_probe/sc; } > $O/r03_stream_ceiling2.txt 2>&1
{
  for n in full nt ntl nts nl nl_nt; do echo "## k_update16 variant $n, 131072 x 512 tiles"; MB_ONLY=512,3,1 $R/tools/_probe/mbu16r3_$n 131072 512 256; MB_ONLY=768,3,1 $R/tools/_probe/mbu16r3_$n 131072 512 256; done
  for n in full nt; do echo "## k_update16 variant $n, 262144 x 256 tiles"; MB_ONLY=512,3,1 $R/tools/_probe/mbu16r3_$n 262144 256 256; done
  for n in full nt; do echo "## k_update16 variant $n, 65536 x 256 tiles (0.25 GiB: the size of a mid-solve 65536^2 pass, resident in the Infinity Cache)"; MB_ONLY=512,3,1 $R/tools/_probe/mbu16r3_$n 65536 256 256; done
} > $O/r03_ntpolicy.txt 2>&1
timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum --kernel-include-regex k_update16 --kernel-trace --output-format csv -d $O/r03_tccr_65536 -- python $R/tools/profile_one.py 65536 1 > $O/r03_tccr_65536.log 2>&1
timeout 300 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_64B_sum --kernel-include-regex k_update16 --kernel-trace --output-format csv -d $O/r03_tccw_65536 -- python $R/tools/profile_one.py 65536 1 > $O/r03_tccw_65536.log 2>&1
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-include-regex k_update16 --kernel-trace --output-format csv -d $O/r03_tcch_65536 -- python $R/tools/profile_one.py 65536 1 > $O/r03_tcch_65536.log 2>&1
timeout 400 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum --kernel-include-regex k_update16 --kernel-iteration-range "[1-32]" --kernel-trace --output-format csv -d $O/r03_tccr_262144 -- python $R/tools/profile_one.py 262144 1 > $O/r03_tccr_262144.log 2>&1
timeout 400 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_64B_sum --kernel-include-regex k_update16 --kernel-iteration-range "[1-32]" --kernel-trace --output-format csv -d $O/r03_tccw_262144 -- python $R/tools/profile_one.py 262144 1 > $O/r03_tccw_262144.log 2>&1
timeout 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-include-regex k_update16 --kernel-iteration-range "[1-32]" --kernel-trace --output-format csv -d $O/r03_tcch_262144 -- python $R/tools/profile_one.py 262144 1 > $O/r03_tcch_262144.log 2>&1
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-include-regex k_update16 --kernel-iteration-range "[1-32]" --kernel-trace --output-format csv -d $O/r03_sqa_262144 -- python $R/tools/profile_one.py 262144 1 > $O/r03_sqa_262144.log 2>&1
timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE --kernel-include-regex k_update16 --kernel-iteration-range "[1-32]" --kernel-trace --output-format csv -d $O/r03_sqb_262144 -- python $R/tools/profile_one.py 262144 1 > $O/r03_sqb_262144.log 2>&1
cd $R; timeout 600 python -m pytest tests -m gpu -x -q -k "devices or bench or batched" > $O/r03_pytest2.log 2>&1; echo "pytest rc=$?" >> $O/r03_pytest2.log
