#!/bin/bash
# Closing session, part A (GPU suite, every bench config with live PMC traffic, whole-call probes): GPU suite (ship + developer build), every bench config with live PMC traffic,
# usage: tools/gpu_session_final.sh <tag>   (tag = r05 ...: every output lands in gpurun_out/<tag>_*; copy what should be judged into profiles/)
# rocprofv3 kernel-trace / PMC summaries, per-pose probes, set_target side, seam probes, soak, rare-event trace, 2-rank bench.
cd "$(dirname "$0")/.."; TAG=${1:-r06}
o=gpurun_out; mkdir -p $o; export TMPDIR=/tmp
timeout 3000 python -m pytest tests -m gpu -x -q -rs --durations=40 > $o/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $o/${TAG}_pytest_gpu.log; tail -4 $o/${TAG}_pytest_gpu.log
for c in plane_b01 icp_b01 icp_b01_harness plane_b01_100k vplane_b01_harness ndt_b01_harness vplane_10m ndt_10m plane_b01_resampled plane_b01_crop plane_lidar icp_lidar_harness plane_100m plane_100m_resampled; do
    timeout 1500 python bench.py --config $c > $o/${TAG}_bench_$c.json 2> $o/${TAG}_bench_$c.err
    python - "$o/${TAG}_bench_$c.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(d["config"]["workload"], "value", d["value"], "ms/step", d["ms_per_step"], "[", d["ms_per_step_min"], d["ms_per_step_max"], "] noev", d["ms_per_step_events_off"],
          {k: v["avg_ms"] for k, v in d["kernels"].items()}, "frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"], "live", (d["roofline"]["traffic_source"] or {}).get("live"),
          "cpu", d.get("cpu_baseline", {}).get("value"), "cores", d.get("cpu_baseline", {}).get("cores"), "seam", d.get("seam"))
except Exception as e:
    print("bench parse failed", sys.argv[1], e)
PY
done
PCR_BENCH_GROUP_DEVICES=0,0 GPU_MAX_HW_QUEUES=8 timeout 600 python bench.py --gpus 2 --single-process --no-pmc --no-cpu-baseline --config plane_b01 --steps 20 --warmup 5 > $o/${TAG}_bench_plane_b01_group2_1gpu.json 2> $o/${TAG}_bench_group2.err; cut -c1-400 $o/${TAG}_bench_plane_b01_group2_1gpu.json
timeout 600 python bench.py --gpus 2 --backend gloo --config plane_b01 --steps 20 --warmup 5 > $o/${TAG}_bench_plane_b01_2ranks_1gpu.json 2> $o/${TAG}_bench_2ranks.err; cut -c1-400 $o/${TAG}_bench_plane_b01_2ranks_1gpu.json
timeout 600 python tools/build_time.py 1.06e6 1e7 1e8 > $o/${TAG}_build_time.txt 2>&1; tail -3 $o/${TAG}_build_time.txt
timeout 600 python tools/speed_test_comparison.py > $o/${TAG}_speed_test_comparison.txt 2>&1; tail -8 $o/${TAG}_speed_test_comparison.txt
timeout 300 python tools/set_target_probe.py 2>&1 | grep -v "^/opt" | head -8 > $o/${TAG}_seam_align.txt; timeout 200 python tools/align_seam_probe.py 2>&1 | grep -v '^/opt' | head -8 >> $o/${TAG}_seam_align.txt; cat $o/${TAG}_seam_align.txt
for c in plane_b01 icp_b01 plane_b01_resampled plane_b01_crop plane_lidar vplane_10m ndt_10m; do
  timeout 900 python tools/reuse_probe.py --config $c --reps 6 --modes 0,1 2>&1 | grep -v "^/opt" > $o/${TAG}_reuse_probe_$c.txt; grep "align" $o/${TAG}_reuse_probe_$c.txt | head -3
done
timeout 1200 python tools/reuse_probe.py --config plane_100m --reps 3 --modes 0,1 --tol 1e-3 2>&1 | grep -v "^/opt" > $o/${TAG}_reuse_probe_plane_100m.txt; grep "align\|trajectory" $o/${TAG}_reuse_probe_plane_100m.txt | head -4
