#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd /tmp
{ for shape in "98304 512" "131072 512" "131072 1024" "262144 512"; do for ord in 0 1; do echo "## rows x tiles = $shape, order $ord"; MB_QUICK=1 MB_ORDER=$ord MB_REPS=4 $R/tools/_probe/mbk $shape | grep "S="; done; done; } > $O/r03_kloop3.txt 2>&1
