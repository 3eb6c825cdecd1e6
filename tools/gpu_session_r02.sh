#!/bin/bash
# One GPU session of round 2: full GPU test suite, every bench config, rocprofv3 passes.
cd "$(dirname "$0")/.."
o=gpurun_out
python -m pytest tests -m gpu -x -q > $o/s_tests.log 2>&1; tail -3 $o/s_tests.log
for c in plane_b01 icp_b01 icp_b01_harness plane_b01_100k vplane_10m ndt_10m plane_100m; do
    python bench.py --config $c > $o/s_bench_$c.json 2> $o/s_bench_$c.err
    python - "$o/s_bench_$c.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(d["config"]["workload"], "value", d["value"], "ms/step", d["ms_per_step"], "[", d["ms_per_step_min"], d["ms_per_step_max"], "] noev", d["ms_per_step_events_off"],
          {k: v["avg_ms"] for k, v in d["kernels"].items()}, "frac", d["roofline"]["frac"], "cpu", d.get("cpu_baseline", {}).get("value"))
except Exception as e:
    print("bench parse failed", sys.argv[1], e)
PY
done
python tools/build_time.py 1.06e6 1e7 1e8 > $o/s_build_time.log 2>&1; tail -3 $o/s_build_time.log
python tools/speed_test_comparison.py > $o/s_speed_test.log 2>&1; tail -8 $o/s_speed_test.log
for c in plane_b01 icp_b01_harness plane_100m; do python tools/pose_profile.py --config $c --modes 0 --reps 10 > $o/s_pose_$c.log 2>&1; grep -h "mean\|walk\|align" $o/s_pose_$c.log; done
tools/collect_profiles.sh r02_plane_b01 plane_b01
tools/collect_profiles.sh r02_plane_b01_coop plane_b01 PCR_NN_MODE=2
tools/collect_profiles.sh r02_plane_100m plane_100m
tools/collect_profiles.sh r02_icp_b01_harness icp_b01_harness
