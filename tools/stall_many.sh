#!/bin/bash
# Run ON THE GPU BOX: the stalls live in the first passes of a PROCESS -> many fresh processes, count the ones with a
# pass > 1 ms, with the host libraries' thread pools at their defaults and capped.
cd /root/repo; export TMPDIR=/tmp
for cap in ${CAPS:-default 1}; do
  n=0; tot=0
  for i in $(seq 1 ${RUNS:-25}); do
    if [ "$cap" = "default" ]; then
      out=$(PCR_STALL_DEBUG=1 python tools/stall_study.py ${PASSES:-3000} small 2>&1 | grep "stall\]\|slow passes\|cgroup")
    else
      out=$(OMP_NUM_THREADS=$cap OPENBLAS_NUM_THREADS=$cap MKL_NUM_THREADS=$cap PCR_STALL_DEBUG=1 python tools/stall_study.py ${PASSES:-3000} small 2>&1 | grep "stall\]\|slow passes\|cgroup")
    fi
    tot=$((tot+1)); if ! echo "$out" | grep -q "\[\]"; then n=$((n+1)); echo "  run $i: $(echo "$out" | tr '\n' ' ' | cut -c1-330)"; fi
  done
  echo "host thread pools: $cap -> $n of $tot fresh processes had a pass > 1 ms"
done
echo "--- one process, ${LONG:-1000000} consecutive unprofiled passes, thread pools capped at 1"
OMP_NUM_THREADS=1 OPENBLAS_NUM_THREADS=1 MKL_NUM_THREADS=1 PCR_STALL_DEBUG=1 python tools/stall_study.py ${LONG:-1000000} small 2>&1 | grep "stall\]\|slow passes\|cgroup\|median"
