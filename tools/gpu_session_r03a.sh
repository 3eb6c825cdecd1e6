#!/bin/bash
# round 3, GPU session: certified reuse -- parity subset, then the probe on the workloads
mkdir -p gpurun_out
cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "certified_reuse or masked_multivoxel or profile_counters or align_loop_edge or shipped_pipeline or scan_order or pure_in or align_matches" > gpurun_out/r03a_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03a_pytest.log
tail -5 gpurun_out/r03a_pytest.log
for c in plane_b01 plane_b01_resampled plane_b01_crop; do
  timeout 600 python tools/reuse_probe.py --config $c --reps 10 > gpurun_out/r03a_probe_$c.txt 2>&1
done
timeout 600 python tools/reuse_probe.py --config vplane_10m --reps 4 > gpurun_out/r03a_probe_vplane_10m.txt 2>&1
timeout 900 python tools/reuse_probe.py --config plane_100m --reps 3 --tol 1e-3 > gpurun_out/r03a_probe_plane_100m.txt 2>&1
grep -v "^/opt" gpurun_out/r03a_probe_plane_b01.txt | tail -30
