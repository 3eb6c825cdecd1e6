#!/bin/bash
# round 6, session n: the small-scan (fused kernel) configs, round-5 library against the current one on one box
cd "$(dirname "$0")/../.."; root=$(pwd); o=$root/gpurun_out; mkdir -p $o; export TMPDIR=/tmp
for rep in 1 2; do
for cfg in icp_b01_harness vplane_b01_harness ndt_b01_harness plane_b01_100k; do
  for lib in r05 ship; do
    if [ $lib = r05 ]; then export PCR_LIB=$root/build/exp/libpcr_r05.so; else unset PCR_LIB; fi
    timeout 600 python bench.py --config $cfg --no-pmc --no-cpu-baseline --repeats 5 2> $o/r06n_$cfg.$lib.err | tail -1 > $o/r06n_bench_$cfg.$lib.json
    python - $o/r06n_bench_$cfg.$lib.json $lib <<'PY' | tee -a $o/r06n_harness_ab.txt
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], d["config"]["workload"], "value", d["value"], "ms/step", d["ms_per_step"], d["repeat_ms_per_step"], "noev", d["ms_per_step_events_off"], {k: v["avg_ms"] for k, v in d["kernels"].items()}, "align", d.get("seam", {}).get("align_ms"))
except Exception as e:
    print(sys.argv[2], "failed", e, open(sys.argv[1].replace("bench_", "").replace(".json", ".err")).read()[-400:])
PY
  done
done
done
