#!/bin/bash
# what the driver runs at round end, on the final build: smoke, the default bench line; plus the bench command's kernel stats
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/r03_smoke55.log 2>&1; echo "smoke rc=$?" > $O/r03_final55.summary
python bench.py > $O/r03_bench_default.json 2> $O/r03_bench_default.err; echo "bench rc=$?" >> $O/r03_final55.summary
cd /tmp && export TMPDIR=/tmp
rm -rf $O/r03_stats_bench
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r03_stats_bench -- python $R/bench.py --no-cpu-baseline --no-batch-c4 --target-n 0 > $O/r03_stats_bench.json 2> $O/r03_stats_bench.err
cd $R; find $O/r03_stats_bench -name "*kernel_trace.csv" -delete
