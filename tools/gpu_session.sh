#!/bin/bash
# ONE parameterised GPU-box session (replaces the per-experiment gpu_session_r04[a-s].sh of round 4).
#   tools/gpu_session.sh <tag> <step> [<step> ...]      steps run in order, everything lands in gpurun_out/<tag>_*.
# steps:
#   tests            the whole -m gpu suite
#   quick            exactness subset (fuzz / stress / reuse / knn)
#   bench:<cfg>      python bench.py --config <cfg>     (bench:default = no --config)
#   poses:<cfg>      per-pose kernel times along the config's trajectory (tools/reuse_probe.py, plain search)
#   pmc:<cfg>:<pose>:<C1,C2,..>   rocprofv3 --pmc counters, 40 passes at one pose (cfg = b01 | 100m)
#   lib:<path>       switch PCR_LIB for the following steps (lib:- = the shipped library)
#   sh:<script>      run another script of tools/ (escape hatch for one-off probes)
root=$(cd "$(dirname "$0")/.." && pwd); out=$root/gpurun_out; mkdir -p $out; export TMPDIR=/tmp
tag=$1; shift
for step in "$@"; do
  IFS=: read -r kind a b c <<< "$step"
  echo "=== [$tag] $step (PCR_LIB=${PCR_LIB:-shipped})"
  case $kind in
    tests) (cd $root && timeout 1500 python -m pytest tests -m gpu -x -q -rs --durations=40 > $out/${tag}_pytest.log 2>&1; echo "rc=$?" >> $out/${tag}_pytest.log; tail -4 $out/${tag}_pytest.log) ;;
    quick) (cd $root && timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fuzz_against_oracle or nn_stress or nn_query or certified_reuse or fuzz_knn or centroid_filter" > $out/${tag}_quick.log 2>&1; echo "rc=$?" >> $out/${tag}_quick.log; tail -3 $out/${tag}_quick.log) ;;
    bench) cfg=""; [ "$a" != default ] && cfg="--config $a"
           (cd $root && timeout 1200 python bench.py $cfg ${BENCH_ARGS:-} 2> $out/${tag}_bench_$a.err | tail -1 | tee $out/${tag}_bench_$a.json) ;;
    poses) (cd $root && timeout 900 python tools/reuse_probe.py --config $a --reps ${REPS:-10} --modes 0 --tol 1e-3 2>&1 | grep -v "^/opt" | tee $out/${tag}_poses_$a.txt) ;;
    pmc)   big=""; [ "$a" = 100m ] && big=100m
           (cd /tmp && rm -rf $out/prof_pp && timeout -k 5 900 rocprofv3 --pmc ${c//,/ } --kernel-trace --output-format rocpd -d $out/prof_pp -o r -- python $root/tools/pose0_passes.py $b $big > $out/prof_pp.log 2>&1
            db=$(find $out/prof_pp -name "*.db" | head -1)
            { echo "== $a pose $b ($c) lib=${PCR_LIB:-shipped}"; python $root/tools/rocpd_last.py "$db" k_nn_scan 40; python $root/tools/rocpd_last.py "$db" k_reduce_finalize 40; } 2>&1 | tee -a $out/${tag}_pmc.txt
            rm -rf $out/prof_pp) ;;
    lib)   if [ "$a" = - ]; then unset PCR_LIB; else export PCR_LIB=$root/$a; fi ;;
    sh)    bash $root/tools/$a ;;
    *) echo "unknown step $step" ;;
  esac
done
