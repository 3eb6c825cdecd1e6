#!/bin/bash
# Round 4, session f: GPU suite on the ship + dev libraries; box-search fix A/B; set_target after the k-NN batching; clean per-pose counters at 1e8; default bench with live PMC.
cd "$(dirname "$0")/.."
o=gpurun_out; mkdir -p $o; export TMPDIR=/tmp
timeout 2700 python -m pytest tests -m gpu -x -q > $o/r04f_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $o/r04f_pytest_gpu.log; tail -4 $o/r04f_pytest_gpu.log
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
done 2>&1 | tee $o/r04f_ab_boxfix.txt
unset PCR_LIB
timeout 300 python tools/set_target_probe.py 2>&1 | grep -v "^/opt" | head -8 | tee $o/r04f_set_target_probe.txt
timeout 300 python tools/set_target_profile.py 1.06e6 12 2>/dev/null | tail -1 | tee -a $o/r04f_set_target_probe.txt
timeout 300 python tools/set_target_profile.py 1e7 6 2>/dev/null | tail -1 | tee -a $o/r04f_set_target_probe.txt
timeout 300 python tools/speed_test_comparison.py 2>&1 | grep -v "^/opt" | tail -8 | tee $o/r04f_speed_test_comparison.txt
unset PCR_BENCH_NO_PMC
timeout 900 python bench.py --no-cpu-baseline > $o/r04f_bench_plane_b01_livepmc.json 2> $o/r04f_bench_plane_b01_livepmc.err; python -c "
import json
d=json.loads(open('$o/r04f_bench_plane_b01_livepmc.json').read().strip().splitlines()[-1])
print('plane_b01 value', d['value'], 'ms', d['ms_per_step'], 'traffic', d['roofline']['traffic'], str(d['roofline']['traffic_source'])[:600])
"
tools/fetch_per_pose_100m.sh "0 12 25" > $o/r04f_fetch100m.log 2>&1; tail -3 $o/r04f_fetch100m.log
