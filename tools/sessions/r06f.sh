#!/bin/bash
# round 6, session f: 16-byte boxes -- exactness (heavy fuzz, lidar, g11), lidar per-pose times and work counters, then the whole GPU suite
cd "$(dirname "$0")/../.."; root=$(pwd); o=$root/gpurun_out; mkdir -p $o; export TMPDIR=/tmp
S=$root/tools/gpu_session.sh
(cd $root && timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -q -rs -k "lidar or g11 or heavy_index or q6_fallback or internal_flag" > $o/r06f_new.log 2>&1; echo "rc=$?" >> $o/r06f_new.log; tail -8 $o/r06f_new.log)
REPS=3 timeout 600 $S r06f poses:plane_lidar
(cd $root && PCR_LIB=$root/point_cloud_registration_amd/libpcr_hip_dev.so timeout 600 python tools/lb_counters_probe.py plane_lidar 2>&1 | grep -v "^/opt" | tee $o/r06f_counters.txt)
BENCH_ARGS="--no-pmc --no-cpu-baseline --repeats 3" timeout 400 $S r06f bench:icp_lidar_harness
$S r06f tests
