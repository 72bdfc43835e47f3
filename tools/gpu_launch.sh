#!/bin/bash
# Run a gpurun call (retrying while the pod's GPU slots are busy) and return when it has ended -- or after ~10 minutes, by
# which time its snapshot of the tree has long been taken and the tree may be edited again.
# usage: tools/gpu_launch.sh <timeout s> <script> ; log: gpurun_out/<script name>.call.log
t=$1; s=$2; name=$(basename $s .sh)
mkdir -p gpurun_out
( for try in 1 2 3 4 5 6 7 8; do
    /usr/local/graft/bin/gpurun --timeout $t -- "bash $s" > gpurun_out/$name.call.log 2>&1
    grep -q "status=transient" gpurun_out/$name.call.log || break
    sleep 45
  done ) &
for i in $(seq 1 130); do
  sleep 5
  if grep -q "status=ok\|status=fail\|status=refused\|status=timeout" gpurun_out/$name.call.log 2>/dev/null; then break; fi
done
tail -2 gpurun_out/$name.call.log
