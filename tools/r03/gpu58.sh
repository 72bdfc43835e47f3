#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -m gpu -x -q > $O/r03_pytest58.log 2>&1; echo "full suite rc=$?" > $O/r03_final58.summary
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/r03_smoke58.log 2>&1; echo "smoke rc=$?" >> $O/r03_final58.summary
python bench.py > $O/r03_bench_default.json 2> $O/r03_bench_default.err; echo "bench rc=$?" >> $O/r03_final58.summary
