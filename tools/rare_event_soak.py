#!/usr/bin/env python3
"""one pass in ~1e6 took 25 ms with the host thread spinning on its CPU (no throttling): a GPU-side
event of unknown cause.  This probe runs N passes of the 100 k-point ICP harness pass (42 us each) with the host pools
capped, records every pass slower than 2 ms with the host's view (wall, thread CPU time), and -- when run under
`rocprofv3 --kernel-trace --output-format rocpd` -- tools/rare_event_report.py finds the matching dispatches (kernel
duration, gap to the previous dispatch) in the trace.     python tools/rare_event_soak.py [passes]"""
import os, sys, time, gc
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1"); os.environ.setdefault("OMP_NUM_THREADS", "1")
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from point_cloud_registration_amd import _capi
from point_cloud_registration_amd.synthetic import street, harness_scan
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
ctx = _capi.get_context(0)
target = street(1_060_000, seed=0); scan = harness_scan(target, 100_000, seed=1)
tgt = _capi.Target.points(ctx, target); sc = _capi.Scan(ctx, scan)
T, it, tr = _capi.align(tgt, sc, _capi.ICP, np.eye(4), 30, 1e-3, 2.0, want_trace=True)
traj = [tr[i, :16].reshape(4, 4).copy() for i in range(it)]
gc.collect(); gc.disable()
slow = []
t_start = time.perf_counter()
for k in range(n):
    c0 = time.thread_time(); t0 = time.perf_counter()
    _capi.linearize(tgt, sc, _capi.ICP, traj[k % len(traj)], 2.0)
    w = time.perf_counter() - t0
    if w > 2e-3:
        slow.append((k, round(w * 1e3, 3), round((time.thread_time() - c0) * 1e3, 3), round(time.perf_counter() - t_start, 3)))
el = time.perf_counter() - t_start
print(f"{n} passes in {el:.1f} s ({el / n * 1e6:.1f} us per pass); passes slower than 2 ms (index, wall ms, thread CPU ms, at s): {slow}")
try:
    print("cpu.stat:", open("/sys/fs/cgroup/cpu.stat").read().replace("\n", " "))
except OSError:
    pass
