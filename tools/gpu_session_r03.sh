#!/bin/bash
# One GPU session of round 3: every bench config, per-pose / reuse probes, build timings, rocprofv3 passes.
cd "$(dirname "$0")/.."
o=gpurun_out; mkdir -p $o; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q > $o/r03_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $o/r03_pytest_gpu.log; tail -3 $o/r03_pytest_gpu.log
for c in plane_b01 icp_b01 icp_b01_harness plane_b01_100k vplane_10m ndt_10m plane_100m plane_b01_resampled plane_b01_crop plane_100m_resampled; do
    timeout 900 python bench.py --config $c > $o/r03_bench_$c.json 2> $o/r03_bench_$c.err
    python - "$o/r03_bench_$c.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(d["config"]["workload"], "value", d["value"], "ms/step", d["ms_per_step"], "[", d["ms_per_step_min"], d["ms_per_step_max"], "] noev", d["ms_per_step_events_off"],
          {k: v["avg_ms"] for k, v in d["kernels"].items()}, "frac", d["roofline"]["frac"], "cpu", d.get("cpu_baseline", {}).get("value"), "seam", d.get("seam"))
except Exception as e:
    print("bench parse failed", sys.argv[1], e)
PY
done
timeout 600 python tools/build_time.py 1.06e6 1e7 1e8 > $o/r03_build_time.txt 2>&1; tail -3 $o/r03_build_time.txt
timeout 600 python tools/speed_test_comparison.py > $o/r03_speed_test_comparison.txt 2>&1; tail -8 $o/r03_speed_test_comparison.txt
for c in plane_b01 plane_b01_resampled plane_b01_crop vplane_10m ndt_10m; do
  timeout 900 python tools/reuse_probe.py --config $c --reps 6 2>&1 | grep -v "^/opt" > $o/r03_reuse_probe_$c.txt; grep "align" $o/r03_reuse_probe_$c.txt
done
timeout 1200 python tools/reuse_probe.py --config plane_100m --reps 3 --tol 1e-3 2>&1 | grep -v "^/opt" > $o/r03_reuse_probe_plane_100m.txt
timeout 1200 python tools/reuse_probe.py --config plane_100m_resampled --reps 3 --modes 0,1 --tol 1e-3 2>&1 | grep -v "^/opt" > $o/r03_reuse_probe_plane_100m_resampled.txt
tools/collect_profiles.sh r03_plane_b01 plane_b01
tools/collect_profiles.sh r03_plane_100m plane_100m
tools/collect_profiles.sh r03_icp_b01_harness icp_b01_harness
tools/collect_profiles.sh r03_vplane_10m vplane_10m
tools/collect_profiles.sh r03_ndt_10m ndt_10m
timeout 200 python tools/align_seam_probe.py 2>&1 | grep -v '^/opt' | head -8 > $o/r03_seam_align.txt; timeout 200 python tools/set_target_probe.py 2>&1 | grep -v '^/opt' | head -8 >> $o/r03_seam_align.txt; cat $o/r03_seam_align.txt
