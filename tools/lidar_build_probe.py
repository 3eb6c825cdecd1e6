import sys, time
sys.path.insert(0, "/root/repo")
from point_cloud_registration_amd import _capi
from point_cloud_registration_amd.synthetic import lidar_sweep
ctx = _capi.get_context(0)
pts = lidar_sweep(1_060_000, 0)
for r in range(8):
    time.sleep(0.03)
    t = _capi.Target.points(ctx, pts); ctx.synchronize(); t.close()
