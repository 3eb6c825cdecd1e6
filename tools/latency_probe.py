#!/usr/bin/env python3
"""Per-step host latency distribution of pcr_linearize (developer probe)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "--mine-first" in sys.argv:
    from point_cloud_registration_amd import _capi as _c0
    _c0.lib(); print("loaded libpcr_hip first")
if "--torch" in sys.argv or "--torchcuda" in sys.argv:
    import torch
if "--torchcuda" in sys.argv:
    torch.cuda.set_device(0); torch.cuda.synchronize()
from point_cloud_registration_amd import _capi
from point_cloud_registration_amd.synthetic import street, perturbed_scan
target = street(1_060_000, seed=0)
scan, _ = perturbed_scan(target, None)
ctx = _capi.get_context(0)
nrm = np.zeros_like(target); nrm[:, 2] = 1
tgt = _capi.Target.points(ctx, target, nrm)
sc = _capi.Scan(ctx, scan)
T = np.eye(4)
if "--nogc" in sys.argv:
    import gc; gc.collect(); gc.disable()
if "--sleep" in sys.argv:
    time.sleep(2.0)
import ctypes
hip = ctypes.CDLL("libamdhip64.so.7")
if "--devsync" in sys.argv:
    print("hipDeviceSynchronize ->", hip.hipDeviceSynchronize())
if "--stream2" in sys.argv:
    st = ctypes.c_void_p(); print("hipStreamCreate ->", hip.hipStreamCreate(ctypes.byref(st)))
if "--malloc" in sys.argv:
    pp = ctypes.c_void_p(); print("hipMalloc ->", hip.hipMalloc(ctypes.byref(pp), ctypes.c_size_t(1 << 20)))
    print("hipMemset(null stream) ->", hip.hipMemset(pp, 0, ctypes.c_size_t(1 << 20)))
os.system(f"grep -E 'libamdhip64|librccl' /proc/{os.getpid()}/maps | awk '{{print $6}}' | sort -u")
for prof in (False,):
    ctx.profile_enable(prof)
    for _ in range(5): _capi.linearize(tgt, sc, 1, T, 2.0)
    ts = []
    tstart = time.perf_counter()
    for _ in range(1500):
        t0 = time.perf_counter(); _capi.linearize(tgt, sc, 1, T, 2.0); ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e3
    big = np.nonzero(ts > 5)[0]
    print("   stalls >5ms at steps", big.tolist(), "ms", ts[big].round(1).tolist(), "t(s)", (np.cumsum(ts)[big] / 1e3).round(3).tolist())
    print(f"prof={prof} env_int={os.environ.get('HSA_ENABLE_INTERRUPT')} torch={'--torch' in sys.argv}: "
          f"min {ts.min():.3f} med {np.median(ts):.3f} p90 {np.percentile(ts,90):.3f} max {ts.max():.3f} mean {ts.mean():.3f}")
    if prof:
        print("   ", ctx.profile_read())
