#!/bin/bash
# every bench config once (run on the GPU box); JSON lines into gpurun_out/s_bench_<config>.json
cd "$(dirname "$0")/.."
for c in plane_b01 icp_b01 icp_b01_harness plane_b01_100k vplane_10m ndt_10m plane_100m; do
    python bench.py --config $c > gpurun_out/s_bench_$c.json 2> gpurun_out/s_bench_$c.err
    python - "gpurun_out/s_bench_$c.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["config"]["workload"], "value", d["value"], "ms/step", d["ms_per_step"], "[", d["ms_per_step_min"], d["ms_per_step_max"], "] noev", d["ms_per_step_events_off"],
      {k: v["avg_ms"] for k, v in d["kernels"].items()}, "frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"], "cpu", d.get("cpu_baseline", {}).get("value"))
PY
done
