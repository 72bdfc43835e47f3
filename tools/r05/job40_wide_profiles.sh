#!/bin/bash
# Late round 5: the outer pass on sixteen wavefronts (k_update16k_wide) -- kernel stats of the large legs, SQ counters and HBM
# traffic of the outer pass under the new default and under the shape of rounds 3-5 (GF2BV_OUTER_SHAPE=1), the isolated kernel,
# then the bench line.  Outputs under gpurun_out/ (copied to profiles/ by hand).
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
SEED=1242 bash tools/jobs/kernel_stats.sh r05w_262144 python tools/profile_one.py 262144 1
bash tools/jobs/kernel_stats.sh r05w_131072 python tools/profile_one.py 131072 1
sq() {  # tag shape
  tag=$1; shape=$2
  cd /tmp && export TMPDIR=/tmp
  i=0
  for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES"; do
    i=$((i+1))
    GF2BV_OUTER_SHAPE=$shape SEED=1242 timeout 600 rocprofv3 --pmc $grp --kernel-include-regex "k_update16k" --kernel-iteration-range "[1-6]" --kernel-trace --output-format csv -d $O/${tag}_sq$i -- python $R/tools/profile_one.py 262144 1 > $O/${tag}_sq$i.log 2>&1
  done
  cd $R
  { echo "# GF2BV_OUTER_SHAPE=$shape python tools/profile_one.py 262144 1 (seed 1242); k_update16k launches [1-6]"; for j in 1 2 3; do python tools/pmc_summary.py $O/${tag}_sq$j; done; } > $O/${tag}_sq.txt 2>&1
  for j in 1 2 3; do find $O/${tag}_sq$j -name "*.csv" -delete; done
}
sq r05w_wide 0
sq r05w_legacy 1
SEED=1242 GF2BV_OUTER_SHAPE=0 bash tools/jobs/pmc_traffic.sh r05w_wide_k16k "k_update16k" --range "[1-6]" -- python tools/profile_one.py 262144 1
for v in 16_512_8 8_1024_2; do tools/_probe/mbk_$v 131072 1024 12; done > $O/r05w_isolated.txt 2>&1
python bench.py > $O/r05w_bench_default.json 2> $O/r05w_bench_default.err
tail -c 1500 $O/r05w_bench_default.json
cat $O/r05w_wide_sq.txt $O/r05w_legacy_sq.txt
