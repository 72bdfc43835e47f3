#!/bin/bash
# Launch a gpurun call in the background and return once its snapshot of the tree has been taken (the call is "running"):
# the tree may be edited again from then on.  usage: tools/gpu_launch.sh <timeout s> <script> ; result: gpurun_out/<script>.call.log
t=$1; s=$2; name=$(basename $s .sh)
mkdir -p gpurun_out
nohup /usr/local/graft/bin/gpurun --timeout $t -- "bash $s" > gpurun_out/$name.call.log 2>&1 &
echo "pid $!"
for i in $(seq 1 120); do
  sleep 5
  if grep -q "push .* in\|status=" gpurun_out/$name.call.log 2>/dev/null; then break; fi
  st=$(/usr/local/graft/bin/gpurun --status 2>/dev/null | tr -d '\n')
  case "$st" in *'"phase": "run'*|*'"state": "run'*|*running*) break;; esac
done
/usr/local/graft/bin/gpurun --status 2>/dev/null | tail -8
