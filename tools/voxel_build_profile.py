#!/usr/bin/env python3
"""Developer probe for rocprofv3: one voxel-target build (1.06 M points, voxel 1.0) + host-side phase timings."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from point_cloud_registration_amd import _capi
from point_cloud_registration_amd.synthetic import street
ctx = _capi.get_context(0)
pts = street(1_060_000)
_capi.Target.voxels(ctx, pts[:100000], 1.0, 10).close()
for rep in range(3):
    t0 = time.perf_counter(); v = _capi.Target.voxels(ctx, pts, 1.0, 10); ctx.synchronize(); t1 = time.perf_counter()
    print(f"voxel build {1e3 * (t1 - t0):.2f} ms ({v.size()} voxels)")
    v.close()
