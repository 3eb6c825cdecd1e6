#!/bin/bash
# chunk size of the interleaved hand-out (tiles per chunk; shipped = 16)
root=$(cd "$(dirname "$0")/../.." && pwd); out=$root/gpurun_out; cd $root
for lib in shipped chunk4 chunk8 chunk32 chunk64; do
  [ $lib = shipped ] && unset PCR_LIB || export PCR_LIB=$root/build/exp/libpcr_$lib.so
  for cfg in plane_b01 icp_b01 plane_100m; do
  echo "== $lib $cfg: nn us per pose"
  timeout 900 python tools/reuse_probe.py --config $cfg --reps $([ $cfg = plane_100m ] && echo 2 || echo 6) --modes 0 --tol 1e-3 2>&1 | grep "pose\|trajectory total" | awk '{ if ($1=="pose") printf "%s ", $14; else print }'
  done
done 2>&1 | tee $out/r05n_chunk.txt
