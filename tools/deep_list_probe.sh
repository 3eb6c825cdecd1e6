#!/bin/bash
# Run ON THE GPU BOX: counters of k_nn_scan at ONE pose of plane_b01 against the halo margin of the ring-0 lists
# (0.1 = shipped, 1.0 = a cell's list holds all 27 cells of its block).   tools/deep_list_probe.sh "0.1 1.0" "0 1 4"
root=$(cd "$(dirname "$0")/.." && pwd); out=$root/gpurun_out; export TMPDIR=/tmp; cd /tmp
: > $out/r04_deep_list_counters.txt
runp() { local halo=$1 pose=$2; shift 2
  rm -rf "$out/prof_dl"
  PCR_HALO=$halo timeout -k 5 200 rocprofv3 --pmc "$@" --kernel-trace --output-format rocpd -d "$out/prof_dl" -o r -- python $root/tools/pose0_passes.py $pose > "$out/prof_dl.log" 2>&1
  local db=$(find "$out/prof_dl" -name "*.db" | head -1)
  echo "== halo $halo pose $pose" >> $out/r04_deep_list_counters.txt
  python "$root/tools/rocpd_summary.py" "$db" 2>&1 | grep -E "k_nn_scan" | grep -v -E "^void.* [0-9]+ +[0-9.]+ +[0-9.]+ +[0-9.]+$" >> $out/r04_deep_list_counters.txt
  rm -rf "$out/prof_dl"; }
for halo in ${1:-0.1 1.0}; do for pose in ${2:-0 4}; do
  runp $halo $pose TA_TA_BUSY TA_TOTAL_WAVEFRONTS GRBM_GUI_ACTIVE
  runp $halo $pose TCP_GATE_EN1 TCP_TOTAL_CACHE_ACCESSES TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ
  runp $halo $pose SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
  runp $halo $pose FETCH_SIZE
done; done
cat $out/r04_deep_list_counters.txt
