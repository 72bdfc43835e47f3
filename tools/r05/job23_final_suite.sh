#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout 2700 python -m pytest tests -m gpu -q > $O/r05_pytest_final2.log 2>&1; echo "full suite rc=$?" > $O/r05_final2.summary
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/r05_smoke_final2.log 2>&1; echo "smoke rc=$?" >> $O/r05_final2.summary
python bench.py > $O/r05_bench_final2.json 2> $O/r05_bench_final2.err; echo "bench rc=$?" >> $O/r05_final2.summary
