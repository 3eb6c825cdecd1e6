#!/bin/bash
# round 6, session m: the device-resident loop reads the deeper list set again (LOCAL == 1 launches); phase-split test
cd "$(dirname "$0")/../.."; root=$(pwd); o=$root/gpurun_out; mkdir -p $o; export TMPDIR=/tmp
(cd $root && timeout 900 python -m pytest tests/test_gpu_phase_split.py tests/test_gpu_parity.py -m gpu -x -q -k "phase_split or deeper_list or align_matches or loops_agree or align_loop" 2>&1 | tail -5 | tee $o/r06m_tests.txt)
for cfg in plane_b01 icp_b01 plane_b01_resampled; do
  timeout 300 python tools/phase_split_probe.py --config $cfg 2>&1 | grep "trajectory\|align" | tee -a $o/r06m_align.txt
done
BENCH_ARGS="--no-pmc --no-cpu-baseline" timeout 400 tools/gpu_session.sh r06m bench:default | cut -c1-200
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06m_bench_default.json").read().strip().splitlines()[-1])
print("bench", d["value"], d["ms_per_step"], d["config"].get("first_align_ms"), d.get("seam"))
PY
