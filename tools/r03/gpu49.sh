#!/bin/bash
# final build: HBM-side traffic of k_update16 at 65536^2 on the KiB-aligned slabs, final default bench line, kernel stats of the bench command
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "k_update16<" --kernel-trace --output-format csv -d $O/r03_fetch_65536_al -- python $R/tools/profile_one.py 65536 1 > $O/r03_fetch_65536_al.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "k_update16<" --kernel-trace --output-format csv -d $O/r03_write_65536_al -- python $R/tools/profile_one.py 65536 1 > $O/r03_write_65536_al.log 2>&1
cd $R
{ python tools/pmc_summary.py $O/r03_fetch_65536_al k_update16; python tools/pmc_summary.py $O/r03_write_65536_al k_update16; } > $O/r03_pmc_aligned.txt 2>&1
find $O/r03_fetch_65536_al $O/r03_write_65536_al -name "*.csv" -delete
python bench.py > $O/r03_bench_default.json 2> $O/r03_bench_default.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r03_stats_bench -- python $R/bench.py --no-cpu-baseline --no-batch-c4 --target-n 0 > $O/r03_stats_bench.json 2> $O/r03_stats_bench.err
cd $R; find $O/r03_stats_bench -name "*kernel_trace.csv" -delete
