#!/bin/bash
# Run ON THE GPU BOX: per-pose fabric traffic and L1 counters of k_nn_scan against the 1e8-point target (12.5 M-point scan),
# 40 passes at ONE pose per run (tools/pose0_passes.py <pose> 100m).   tools/fetch_per_pose_100m.sh "0 5 12 20 25"
root=$(cd "$(dirname "$0")/.." && pwd); out=$root/gpurun_out; export TMPDIR=/tmp; cd /tmp
res=$out/r04_plane_100m_fetch_per_pose.txt; : > $res
runp() { local pose=$1; shift
  rm -rf "$out/prof_pp"
  timeout -k 5 600 rocprofv3 --pmc "$@" --kernel-trace --output-format rocpd -d "$out/prof_pp" -o r -- python $root/tools/pose0_passes.py $pose 100m > "$out/prof_pp.log" 2>&1
  local db=$(find "$out/prof_pp" -name "*.db" | head -1)
  echo "== pose $pose ($*)" >> $res
  python "$root/tools/rocpd_last.py" "$db" k_nn_scan 40 >> $res 2>&1
  python "$root/tools/rocpd_last.py" "$db" k_reduce_finalize 40 >> $res 2>&1
  rm -rf "$out/prof_pp"; }
for pose in ${1:-0 5 12 20 25}; do
  runp $pose FETCH_SIZE
  runp $pose TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_GATE_EN1 TA_TA_BUSY
done
cat $res
