#!/bin/bash
# final build: whole GPU suite, the default bench line, and the kernel stats of the bench command (headline leg only)
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -x -q > $O/r03_pytest24.log 2>&1; echo "full suite rc=$?" > $O/r03_pytest24.summary
python bench.py > $O/r03_bench_default.json 2> $O/r03_bench_default.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r03_stats_bench -- python $R/bench.py --no-cpu-baseline --no-batch-c4 --target-n 0 > $O/r03_stats_bench.json 2> $O/r03_stats_bench.err
cd $R; find $O/r03_stats_bench -name "*kernel_trace.csv" -delete
