"""Device-memory leak check: free HBM before/after repeated create/align/destroy cycles."""
import sys, numpy as np, ctypes
sys.path.insert(0, '/root/repo')
from point_cloud_registration_amd import _capi
import point_cloud_registration_amd as pcr
from point_cloud_registration_amd.synthetic import street, perturbed_scan
import torch
def free_mb():
    return torch.cuda.mem_get_info(0)[0] / 2**20
target = street(300_000, seed=0); scan, _ = perturbed_scan(target, 100_000)
ctx = _capi.get_context(0)
def cycle():
    for cls, kw in ((pcr.ICP, {}), (pcr.PlaneICP, {"k": 8}), (pcr.VPlaneICP, {}), (pcr.NDT, {})):
        m = cls(**kw); m.set_target(target); m.align(scan); m.calc_H_g_e2(np.eye(4), scan)
        tree = pcr.KDTree(target); tree.query(scan[:1000]); tree.query(scan[:100], k=5)
        pcr.voxel_filter(target, 0.5)
cycle(); import gc; gc.collect(); ctx.synchronize()
f0 = free_mb()
for i in range(15):
    cycle()
gc.collect(); ctx.synchronize()
f1 = free_mb()
print(f"free before {f0:.1f} MiB, after 15 more cycles {f1:.1f} MiB, delta {f0 - f1:.1f} MiB")
