#!/bin/bash
# round 6, session k: non-temporal nn_j stores A/B (bench lines), then the whole GPU suite with per-test durations
cd "$(dirname "$0")/../.."; root=$(pwd); o=$root/gpurun_out; mkdir -p $o; export TMPDIR=/tmp
for cfg in vplane_10m ndt_10m plane_b01 icp_b01 plane_100m; do
  for lib in nt0 ship; do
    if [ $lib = nt0 ]; then export PCR_LIB=$root/point_cloud_registration_amd/variants/libpcr_hip_nt0.so; else unset PCR_LIB; fi
    timeout 600 python bench.py --config $cfg --no-pmc --no-cpu-baseline --repeats 5 2> $o/r06k_$cfg.$lib.err | tail -1 > $o/r06k_bench_$cfg.$lib.json
    python - $o/r06k_bench_$cfg.$lib.json $lib <<'PY' | tee -a $o/r06k_nt_ab.txt
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], d["config"]["workload"], "value", d["value"], "ms/step", d["ms_per_step"], d["repeat_ms_per_step"], "noev", d["ms_per_step_events_off"], {k: v["avg_ms"] for k, v in d["kernels"].items()}, "align", d.get("seam", {}).get("align_ms"))
PY
  done
done
unset PCR_LIB
tools/gpu_session.sh r06k tests
