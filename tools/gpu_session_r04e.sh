#!/bin/bash
# Round 4, session e (measurement only): set_target-side profiles, deep-list counters at plane_b01, per-pose traffic at 1e8 points.
cd "$(dirname "$0")/.."
o=gpurun_out; mkdir -p $o
tools/collect_set_target_profiles.sh > $o/r04e_set_target.log 2>&1; tail -5 $o/r04e_set_target.log
tools/deep_list_probe.sh "0.1 1.0" "0 4" > $o/r04e_deep_list.log 2>&1; tail -5 $o/r04e_deep_list.log
timeout 300 python tools/set_target_probe.py > $o/r04e_set_target_probe.txt 2>&1; head -8 $o/r04e_set_target_probe.txt
tools/fetch_per_pose_100m.sh "0 5 12 20 25" > $o/r04e_fetch100m.log 2>&1; tail -5 $o/r04e_fetch100m.log
