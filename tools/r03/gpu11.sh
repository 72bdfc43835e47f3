#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd /tmp
{ for b in mbk2 mbk2_s12; do $R/tools/_probe/$b 131072 1024 4; $R/tools/_probe/$b 131072 1024 8; $R/tools/_probe/$b 131072 1024 2; MB_WGS=240 $R/tools/_probe/$b 131072 1024 4; done; } > $O/r03_kloop5.txt 2>&1
