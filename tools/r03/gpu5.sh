#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd $R; timeout 1500 python -m pytest tests -m gpu -x -q > $O/r03_pytest5.log 2>&1; echo "pytest rc=$?" >> $O/r03_pytest5.log
python bench.py --workload sharded --n 65536 --steps 3 --warmup 1 > $O/r03_sharded_65536.json 2> $O/r03_sharded_65536.err
python tools/profile_one.py 65536 3 > $O/r03_plain_65536.txt 2>&1
