#!/bin/bash
# round 6, session d: groups with more hardware queues; batched box scans on the lidar configs (A/B vs the sequential walk, halo off,
# cell sizes); the new reference fixtures (g11, g12), the PCD / single-process bench tests; the parity ledger
cd "$(dirname "$0")/../.."; root=$(pwd); o=$root/gpurun_out; mkdir -p $o; export TMPDIR=/tmp
S=$root/tools/gpu_session.sh
(cd $root && timeout 1500 python -m pytest tests/test_gpu_group.py -m gpu -q -rs > $o/r06d_group.log 2>&1; echo "rc=$?" >> $o/r06d_group.log; tail -5 $o/r06d_group.log)
(cd $root && timeout 1200 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py tests/test_gpu_bench_two_ranks.py -m gpu -q -rs -k "lidar or g11 or voxel_filter or centroid_tree or b01_pcd or single_process" > $o/r06d_new.log 2>&1; echo "rc=$?" >> $o/r06d_new.log; tail -15 $o/r06d_new.log)
REPS=3 timeout 600 $S r06d poses:plane_lidar
PCR_LIB=$root/build/exp/libpcr_lb_seq.so REPS=3 timeout 600 $S r06d_seq poses:plane_lidar
PCR_HALO=0 REPS=3 timeout 600 $S r06d_halo0 poses:plane_lidar
PCR_GRID_CELL=0.2 REPS=3 timeout 600 $S r06d_cell0.2 poses:plane_lidar
PCR_CELLS_PER_POINT=16 REPS=3 timeout 600 $S r06d_cpp16 poses:plane_lidar
PCR_CELLS_PER_POINT=32 REPS=3 timeout 600 $S r06d_cpp32 poses:plane_lidar
BENCH_ARGS="--no-pmc --no-cpu-baseline --repeats 3" timeout 400 $S r06d bench:icp_lidar_harness
(cd $root && timeout 900 python tools/parity_ledger.py > $o/r06_g8_parity.txt 2> $o/r06_g8_parity.err; tail -3 $o/r06_g8_parity.txt)
(cd $root && timeout 600 python -m pytest tests/test_gpu_two_ranks.py -m gpu -q -rs > $o/r06d_ranks.log 2>&1; echo "rc=$?" >> $o/r06d_ranks.log; tail -5 $o/r06d_ranks.log)
