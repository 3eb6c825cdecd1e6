#!/bin/bash
root=$(cd "$(dirname "$0")/../.." && pwd); out=$root/gpurun_out; cd $root
PCR_LIB=$root/build/exp/libpcr_mfstats.so timeout 600 python tools/mf_stats_probe.py plane_b01 2>&1 | grep -v "^/opt" | tee $out/r05g_mf_stats.txt
