#!/bin/bash
# Run ON THE GPU BOX (through gpurun): texture-addresser / L1 (TA, TCP, TD) and SQ utilisation counters of the
# pass kernels on plane_b01, a few counters per rocprofv3 pass (more per pass exceeds the hardware and aborts),
# every pass under its own timeout.  Output: gpurun_out/ta_probe.txt (per-instance averages per launch).
root=$(cd "$(dirname "$0")/.." && pwd); out=$root/gpurun_out; export TMPDIR=/tmp; cd /tmp
rocprofv3 --list-avail 2>/dev/null | grep -o -E "\b(TA_[A-Z_0-9]+|TCP_[A-Z_0-9]+|TD_[A-Z_0-9]+|SQ_INST_CYCLES_VMEM[A-Z_]*|SQ_ACTIVE_INST_[A-Z_]+|SQ_WAIT_INST_[A-Z_]+|SQ_BUSY_CYCLES|SQ_INSTS_VMEM[A-Z_]*|SQ_THREAD_CYCLES_VALU|SQ_INST_LEVEL_VMEM)\b" | sort -u | tr "\n" " " > $out/avail_counters.txt
cmd="python $root/bench.py --config plane_b01 --steps 20 --warmup 5 --repeats 2 --no-cpu-baseline"
run() { local name=$1; shift
  rm -rf "$out/prof_ta_$name"
  timeout -k 5 150 rocprofv3 --pmc "$@" --kernel-trace --output-format rocpd -d "$out/prof_ta_$name" -o r -- $cmd > "$out/prof_ta_$name.log" 2>&1
  local db=$(find "$out/prof_ta_$name" -name "*.db" | head -1)
  python "$root/tools/rocpd_summary.py" "$db" 2>&1 | grep -E "k_nn_scan|k_reduce_fin" | grep -v -E "^void.* [0-9]+ +[0-9.]+ +[0-9.]+ +[0-9.]+$" >> $out/ta_probe.txt
  rm -rf "$out/prof_ta_$name"; }
: > $out/ta_probe.txt
run a TA_TA_BUSY TA_TOTAL_WAVEFRONTS GRBM_GUI_ACTIVE
run b TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES
run c TCP_GATE_EN1 TCP_GATE_EN2 TCP_TCP_TA_DATA_STALL_CYCLES TCP_PENDING_STALL_CYCLES
run d TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_READ_TAGCONFLICT_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES
run e SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INST_LEVEL_VMEM SQ_WAVE_CYCLES
run f TD_TD_BUSY TD_TC_STALL TD_LOAD_WAVEFRONT
cat $out/ta_probe.txt
# the same per pose: 40 passes at the first / the converged pose of the trajectory (tools/pose0_passes.py)
: > $out/ta_probe_pose.txt
runp() { local pose=$1; shift; local name=$1; shift
  rm -rf "$out/prof_p_$name"
  timeout -k 5 150 rocprofv3 --pmc "$@" --kernel-trace --output-format rocpd -d "$out/prof_p_$name" -o r -- python $root/tools/pose0_passes.py $pose > "$out/prof_p_$name.log" 2>&1
  local db=$(find "$out/prof_p_$name" -name "*.db" | head -1)
  echo "== pose $pose" >> $out/ta_probe_pose.txt
  python "$root/tools/rocpd_summary.py" "$db" 2>&1 | grep -E "k_nn_scan" | grep -v -E "^void.* [0-9]+ +[0-9.]+ +[0-9.]+ +[0-9.]+$" >> $out/ta_probe_pose.txt
  rm -rf "$out/prof_p_$name"; }
for pose in 0 4; do
  runp $pose a$pose TA_TA_BUSY TA_TOTAL_WAVEFRONTS GRBM_GUI_ACTIVE
  runp $pose b$pose TCP_GATE_EN1 TCP_TOTAL_CACHE_ACCESSES TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ
  runp $pose c$pose SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
done
cat $out/ta_probe_pose.txt
