#!/bin/bash
# chunk-interleaved tile hand-out (PCR_TILE_INTERLEAVE=1) against the contiguous spans (=0): per-pose search times
root=$(cd "$(dirname "$0")/../.." && pwd); out=$root/gpurun_out; cd $root
PCR_TILE_INTERLEAVE=1 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fuzz_against_oracle or nn_stress or linearize_street or certified_reuse or centroid_filter or quirk_q6" 2>&1 | tail -3
for il in 0 1; do
  for cfg in plane_b01 icp_b01 plane_b01_resampled; do
  echo "== PCR_TILE_INTERLEAVE=$il $cfg: nn us per pose"
  PCR_TILE_INTERLEAVE=$il timeout 600 python tools/reuse_probe.py --config $cfg --reps 6 --modes 0 --tol 1e-3 2>&1 | grep "pose\|trajectory total" | awk '{ if ($1=="pose") printf "%s ", $14; else print }'
  done
done 2>&1 | tee $out/r05k2_interleave.txt
for il in 0 1; do
  echo "== PCR_TILE_INTERLEAVE=$il plane_100m"
  PCR_TILE_INTERLEAVE=$il timeout 900 python tools/reuse_probe.py --config plane_100m --reps 2 --modes 0 --tol 1e-3 2>&1 | grep "pose\|trajectory total" | awk '{ if ($1=="pose") printf "%s ", $14; else print }'
done 2>&1 | tee -a $out/r05k2_interleave.txt
