#!/usr/bin/env python3
"""Developer study of the rare 12-42 ms launch stalls of the zero-copy hand-off: N unprofiled
passes, per-pass wall time, indices and sizes of the slow ones (their spacing in LAUNCHES points at the cause).

    PCR_RETIRE_PERIOD=0|8|64 python tools/stall_study.py [passes] [config: small|large]
"""
import os, sys, time, gc
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from point_cloud_registration_amd import _capi
from point_cloud_registration_amd.synthetic import street, harness_scan, perturbed_scan
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
cfg = sys.argv[2] if len(sys.argv) > 2 else "small"
ctx = _capi.get_context(0)
target = street(1_060_000, seed=0)
scan = harness_scan(target, 100_000, seed=1) if cfg == "small" else perturbed_scan(target, None, seed=2)[0]
tgt = _capi.Target.points(ctx, target); sc = _capi.Scan(ctx, scan)
T, it, tr = _capi.align(tgt, sc, _capi.ICP, np.eye(4), 30, 1e-3, 2.0, want_trace=True)
traj = [tr[i, :16].reshape(4, 4).copy() for i in range(it)]
launches_per_pass = 1 if cfg == "small" else 2
def cpu_stat():
    for path in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
        try:
            return {l.split()[0]: int(l.split()[1]) for l in open(path)}
        except OSError:
            pass
    return {}
cs0 = cpu_stat()
gc.collect(); gc.disable()
ts = np.empty(n)
t_start = time.perf_counter()
t_proc0 = time.time()
for k in range(n):
    t0 = time.perf_counter(); _capi.linearize(tgt, sc, _capi.ICP, traj[k % len(traj)], 2.0); ts[k] = time.perf_counter() - t0
total = time.perf_counter() - t_start
slow = np.nonzero(ts > 1e-3)[0]
print(f"retire_period={os.environ.get('PCR_RETIRE_PERIOD', '0')} cfg={cfg} passes={n} ({launches_per_pass} launches each) "
      f"median {np.median(ts) * 1e6:.1f} us  p99.9 {np.quantile(ts, 0.999) * 1e6:.1f} us  max {ts.max() * 1e3:.2f} ms  total {total:.2f} s")
print("slow passes (> 1 ms): index, ms, seconds into the loop:", [(int(i), round(float(ts[i]) * 1e3, 2), round(float(ts[:i].sum()), 4)) for i in slow[:40]])
cs1 = cpu_stat()
print("cgroup cpu.stat deltas over the loop:", {k: cs1[k] - cs0[k] for k in cs1 if "thrott" in k or k == "nr_periods"},
      "| cpu.max:", (open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "n/a"),
      "| affinity:", len(os.sched_getaffinity(0)), "cpus")
try:
    import psutil
    print("process age at loop start: %.2f s" % (t_proc0 - psutil.Process().create_time()))
except Exception:
    pass
if len(slow) > 1:
    print("spacing between slow passes:", np.diff(slow)[:40].tolist())
