#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
{ echo "# K-loop costing ($(date -u))"; $R/tools/_probe/mbk 98304 512; echo "# product kernel on the same box, same matrix size"; MB_ONLY=512,3,1 $R/tools/_probe/mbu16r3_full 98304 512 256; MB_ONLY=512,3,1 $R/tools/_probe/mbu16_l2 98304 512 256; } > $O/r03_kloop.txt 2>&1
cd $R; timeout 900 python -m pytest tests -m gpu -x -q -k "devices or bench or batched" > $O/r03_pytest3.log 2>&1; echo "pytest rc=$?" >> $O/r03_pytest3.log
