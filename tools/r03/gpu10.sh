#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd /tmp
{ $R/tools/_probe/mbk2 131072 1024 4; MB_ALIGNED=1 $R/tools/_probe/mbk2 131072 1024 4; $R/tools/_probe/mbk2 131072 1024 8; MB_ALIGNED=1 $R/tools/_probe/mbk2 131072 1024 8; MB_WGS=240 $R/tools/_probe/mbk2 131072 1024 4; MB_WGS=224 $R/tools/_probe/mbk2 131072 1024 4; MB_QUICK=1 MB_REPS=6 $R/tools/_probe/mbk 131072 1024 | grep "S="; } > $O/r03_kloop4.txt 2>&1
