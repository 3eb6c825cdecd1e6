#!/bin/bash
# Run ON THE GPU BOX: trace the first passes of a process (where the rare launch stalls live) with rocprofv3's API
# traces (no counters) and list the longest HIP / HSA API calls with what they enclose.
mkdir -p /root/repo/gpurun_out; cd /tmp; export TMPDIR=/tmp
for attempt in 1 2 3 4; do
  rm -rf /tmp/stall_tr
  PCR_RETIRE_PERIOD=${PERIOD:-0} rocprofv3 --hip-trace --hsa-trace --output-format rocpd -d /tmp/stall_tr -o r -- python /root/repo/tools/stall_study.py ${PASSES:-4000} small > /tmp/stall_tr.log 2>&1
  grep "slow passes\|retire_period" /tmp/stall_tr.log
  db=$(find /tmp/stall_tr -name "*.db" | head -1)
  python /root/repo/tools/stall_trace_summary.py "$db" | tee /root/repo/gpurun_out/stall_trace_$attempt.txt | head -40
done
