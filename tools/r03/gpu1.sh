#!/bin/bash
# round 3, first GPU call: stream ceilings (independent tools), 512-pivot costing, baseline suite, SQ / WRITE_SIZE counters
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
{
  echo "# r03 stream ceiling: one box, one run ($(date -u))"; rocm-smi --showclocks 2>/dev/null | grep -i "mclk\|sclk" | head -4
  $R/tools/_probe/sc
  python $R/tools/stream_ceiling_torch.py
} > $O/r03_stream_ceiling.txt 2>&1
{
  echo "# 512 pivots x 8 B (tools/microbench_update8x512.hip), 131072 rows x 1024 one-word tiles = 1 GiB"
  for v in "" -DMB_NOLOOKUP -DMB_L2 -DMB_NOMULT; do $R/tools/_probe/mbu8x$v 131072 1024 256; done
  echo "# same at 262144 rows x 512 tiles (multipliers 16 MiB)"
  for v in "" -DMB_NOLOOKUP; do $R/tools/_probe/mbu8x$v 262144 512 256; done
  echo "# product kernel k_update16, 131072 rows x 512 two-word tiles = 1 GiB: full / stream only / table work only"
  $R/tools/_probe/mbu16_full 131072 512 256; $R/tools/_probe/mbu16_nolookup 131072 512 256; $R/tools/_probe/mbu16_l2 131072 512 256
  echo "# product kernel at 262144 rows x 256 tiles (multipliers 8 MiB)"
  $R/tools/_probe/mbu16_full 262144 256 256; $R/tools/_probe/mbu16_nolookup 262144 256 256
  echo "# K-loop bound: a workgroup can hold 2048 row segments in registers -> tables rebuilt per 2048 rows (rows = 2304: 2048 + the 256 pivot rows)"
  MB_ONLY=512,3,1 $R/tools/_probe/mbu16_l2 2304 16384 256; MB_ONLY=512,3,1 $R/tools/_probe/mbu16_full 2304 16384 256
  MB_ONLY=512,3,1 $R/tools/_probe/mbu16_l2 4352 8192 256
} > $O/r03_costing.txt 2>&1
cd $R; timeout 900 python -m pytest tests -m gpu -x -q > $O/r03_pytest1.log 2>&1; echo "pytest rc=$?" >> $O/r03_pytest1.log
cd /tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-include-regex k_update16 --kernel-trace --output-format csv -d $O/r03_sqa_65536 -- python $R/tools/profile_one.py 65536 1 > $O/r03_sqa_65536.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU GRBM_GUI_ACTIVE --kernel-include-regex k_update16 --kernel-trace --output-format csv -d $O/r03_sqb_65536 -- python $R/tools/profile_one.py 65536 1 > $O/r03_sqb_65536.log 2>&1
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-include-regex k_update16 --kernel-trace --output-format csv -d $O/r03_tcc_65536 -- python $R/tools/profile_one.py 65536 1 > $O/r03_tcc_65536.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex k_update16 --kernel-iteration-range "[1-32]" --kernel-trace --output-format csv -d $O/r03_write_262144 -- python $R/tools/profile_one.py 262144 1 > $O/r03_write_262144.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex k_update16 --kernel-iteration-range "[1-32]" --kernel-trace --output-format csv -d $O/r03_fetch_262144 -- python $R/tools/profile_one.py 262144 1 > $O/r03_fetch_262144.log 2>&1
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-include-regex k_update16 --kernel-iteration-range "[1-32]" --kernel-trace --output-format csv -d $O/r03_tcc_262144 -- python $R/tools/profile_one.py 262144 1 > $O/r03_tcc_262144.log 2>&1
rocprofv3-avail list 2>/dev/null | grep -i "mall\|dram\|EA0_RD\|EA0_WR\|TCC_EA" | head -60 > $O/r03_counters_avail.txt
ls $O
