#!/bin/bash
# Round 4, session c: GPU suite; A/B of the voxel configs (pending points compacted per wave in the reduce prologue);
# per-pose search time of plane_b01 against the halo margin of the ring-0 lists (up to a whole cell).
cd "$(dirname "$0")/.."
o=gpurun_out; mkdir -p $o; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q > $o/r04c_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $o/r04c_pytest_gpu.log; tail -4 $o/r04c_pytest_gpu.log
export PCR_BENCH_NO_RCCL_PROBE=1
for c in vplane_10m ndt_10m; do
  for lib in base new base new; do
    if [ $lib = base ]; then export PCR_LIB=$PWD/build/exp/libpcr_base.so; else unset PCR_LIB; fi
    timeout 600 python bench.py --config $c --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$c $lib', 'ms/step', d['ms_per_step'], 'min', d['ms_per_step_min'], 'noev', d['ms_per_step_events_off'], {k: v['avg_ms'] for k, v in d['kernels'].items()}, 'align', d['seam']['align_ms'])
"
  done
done 2>&1 | tee $o/r04c_ab_fix.txt
unset PCR_LIB
for h in 0.1 0.25 0.45 0.7 1.0; do
  echo "== PCR_HALO=$h"
  PCR_HALO=$h timeout 600 python tools/reuse_probe.py --config plane_b01 --reps 8 --modes 0 --tol 1e-3 2>&1 | grep "pose\|trajectory total\|GN iter"
done 2>&1 | tee $o/r04c_halo_per_pose.txt
for h in 0.1 0.45 1.0; do
  echo "== resampled PCR_HALO=$h"
  PCR_HALO=$h timeout 600 python tools/reuse_probe.py --config plane_b01_resampled --reps 6 --modes 0 --tol 1e-3 2>&1 | grep "pose\|trajectory total\|GN iter"
done 2>&1 | tee -a $o/r04c_halo_per_pose.txt
PCR_HALO=1.0 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fuzz_against or g2 or linearize" 2>&1 | tail -3 | tee $o/r04c_halo1_tests.txt
