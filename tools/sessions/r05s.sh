#!/bin/bash
# hybrid hand-out for large scans: a fraction of every XCD's tiles block-locally, the rest through the counters
root=$(cd "$(dirname "$0")/../.." && pwd); out=$root/gpurun_out; cd $root
PCR_TILE_HYBRID=0.7 timeout 1200 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "10m_voxel_paths or 10m_centroid or g10" 2>&1 | tail -3
for h in 0 0.5 0.75 0.9; do
  for cfg in vplane_10m ndt_10m plane_100m; do
  echo "== PCR_TILE_HYBRID=$h $cfg"
  PCR_TILE_HYBRID=$h timeout 900 python tools/reuse_probe.py --config $cfg --reps $([ $cfg = plane_100m ] && echo 2 || echo 5) --modes 0 --tol 1e-3 2>&1 | grep "pose\|trajectory total" | awk '{ if ($1=="pose") printf "%s ", $14; else print }'
  done
done 2>&1 | tee $out/r05s_hybrid.txt
