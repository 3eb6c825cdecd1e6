#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 passes over the bench command, summaries into gpurun_out/.
#   tools/collect_profiles.sh <tag> <bench-config> [extra env, e.g. PCR_NN_MODE=2]
# Passes (separate runs, as MI355X_MICROARCH.md prescribes for the TCC counters):
#   1. --kernel-trace --stats             -> <tag>_kernel_stats.txt
#   2. --pmc FETCH_SIZE                   -> <tag>_pmc_fetch.txt
#   3. --pmc WRITE_SIZE                   -> <tag>_pmc_write.txt
#   4. --pmc SQ_* (wave cycles, waits, VALU/SALU/LDS instruction counts)   -> <tag>_pmc_sq.txt
set -u
tag=$1; cfg=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out
export TMPDIR=/tmp
for e in "$@"; do export "$e"; done
cmd="python $root/bench.py --config $cfg --steps 20 --warmup 5 --repeats 2 --no-cpu-baseline"
cd /tmp
run() {   # name, rocprofv3 args...
    local name=$1; shift
    rm -rf "$out/prof_${tag}_$name"
    rocprofv3 "$@" --kernel-trace --output-format rocpd -d "$out/prof_${tag}_$name" -o r -- $cmd > "$out/prof_${tag}_$name.log" 2>&1
    local db=$(find "$out/prof_${tag}_$name" -name "*.db" | head -1)
    python "$root/tools/rocpd_summary.py" "$db" > "$out/${tag}_$name.txt" 2>&1
    rm -rf "$out/prof_${tag}_$name"
}
run kernel_stats --stats
run pmc_fetch --pmc FETCH_SIZE
run pmc_write --pmc WRITE_SIZE
run pmc_sq --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES
head -6 "$out/${tag}_kernel_stats.txt"
