#!/bin/bash
# gpurun with retries while the pod has no free GPU slot (exit code 3: nothing charged).  usage: tools/gpurun_retry.sh <timeout_s> '<command>'
t=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $t -- "$@"; rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 90
done
exit 3
