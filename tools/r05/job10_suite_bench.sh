#!/bin/bash
# round 5: full GPU suite + smoke + default bench line on the current build
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 2700 python -m pytest tests -m gpu -q -x > $O/r05_pytest_mid.log 2>&1; echo "full suite rc=$?" > $O/r05_mid.summary
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/r05_smoke_mid.log 2>&1; echo "smoke rc=$?" >> $O/r05_mid.summary
python bench.py > $O/r05_bench_mid.json 2> $O/r05_bench_mid.err; echo "bench rc=$?" >> $O/r05_mid.summary
