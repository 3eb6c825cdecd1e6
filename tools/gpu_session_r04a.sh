#!/bin/bash
# Round 4, session a: the new parity / plumbing tests first, then the whole GPU suite, then the default bench.
cd "$(dirname "$0")/.."
o=gpurun_out; mkdir -p $o; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_bench_two_ranks.py tests/test_gpu_fullsize.py -m gpu -x -q -k "two_ranks or strong or pinned or 10m" > $o/r04a_new_tests.log 2>&1; echo "rc=$?" >> $o/r04a_new_tests.log; tail -15 $o/r04a_new_tests.log
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "q6 or centroid_filter" > $o/r04a_q6.log 2>&1; echo "rc=$?" >> $o/r04a_q6.log; tail -5 $o/r04a_q6.log
timeout 2400 python -m pytest tests -m gpu -x -q > $o/r04a_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $o/r04a_pytest_gpu.log; tail -4 $o/r04a_pytest_gpu.log
timeout 900 python bench.py > $o/r04a_bench_plane_b01.json 2> $o/r04a_bench_plane_b01.err; cat $o/r04a_bench_plane_b01.json | cut -c1-1500
