#!/bin/bash
# Round 4, session g: GPU suite; prologue (unrolled walk + box search) A/B on the voxel configs; adaptive list depth A/B on the point configs.
cd "$(dirname "$0")/.."
o=gpurun_out; mkdir -p $o; export TMPDIR=/tmp
timeout 2700 python -m pytest tests -m gpu -x -q > $o/r04g_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $o/r04g_pytest_gpu.log; tail -4 $o/r04g_pytest_gpu.log
export PCR_BENCH_NO_RCCL_PROBE=1 PCR_BENCH_NO_PMC=1
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', 'ms/step', d['ms_per_step'], 'min', d['ms_per_step_min'], 'noev', d['ms_per_step_events_off'], {k: v['avg_ms'] for k, v in d['kernels'].items()}, 'align', d['seam']['align_ms'], 'class align', d['seam'].get('class_align_from_host_array_ms'), 'set_target', d['seam'].get('set_target_ms'))
"; }
for c in vplane_10m ndt_10m; do
  for v in base new base new; do
    unset PCR_LIB
    [ $v = base ] && export PCR_LIB=$PWD/build/exp/libpcr_base.so
    timeout 600 python bench.py --config $c --no-cpu-baseline 2>/dev/null | line "$c $v"
  done
done 2>&1 | tee $o/r04g_ab_voxel.txt
for c in plane_b01 icp_b01 plane_b01_resampled; do
  for v in prev new prev new; do
    unset PCR_LIB
    [ $v = prev ] && export PCR_LIB=$PWD/build/exp/libpcr_prev.so
    timeout 600 python bench.py --config $c --no-cpu-baseline 2>/dev/null | line "$c $v"
  done
done 2>&1 | tee $o/r04g_ab_halo2.txt
unset PCR_LIB
timeout 600 python tools/reuse_probe.py --config plane_b01 --reps 8 --modes 0 --tol 1e-3 2>&1 | grep "pose\|trajectory total\|GN iter\|align" | tee $o/r04g_pose_plane_b01.txt
tools/fetch_per_pose_100m.sh "0 12 25" > $o/r04g_fetch100m.log 2>&1; cat $o/r04_plane_100m_fetch_per_pose.txt
