"""Worker of tests/test_gpu_group.py (a script, not a test module): the single-process multi-device path on N contexts of
GPU 0 -- ``python tests/group_check.py N``.  Exit code 0 = every check passed.

Checks, for all four kinds on the reference-run fixture g2 (tests/golden/make_golden.py):
  * ``cls(devices=[0] * N)`` with the reference's unchanged call order (set_target(target); align(scan); calc_H_g_e2(T, scan))
    returns the reference's iteration count, its pose to 1e-4 and its H to 1e-5;
  * the group's 29 sums are BIT-IDENTICAL to what the SPMD peer-to-peer run produces: the members' local sums added in
    rank order starting from 0.0 (k_p2p_allreduce) -- recomputed here shard by shard on a plain single context;
  * the device-resident loop of pcr_group_align equals the host loop over pcr_group_linearize bit for bit;
  * 70 consecutive exchanges (the 64-slot table wraps) stay exact.
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

import point_cloud_registration_amd as pcr                       # noqa: E402
from point_cloud_registration_amd import _capi, distributed as pdist      # noqa: E402
from conftest import load_golden                                  # noqa: E402


def main(n):
    devs = [0] * n
    g2 = load_golden("g2_mini_street.npz")
    md, vs = float(g2["max_dist"]), float(g2["voxel_size"])
    target, source = g2["target"], g2["source"]
    ctx = _capi.get_context(0)
    kinds = {"plane": _capi.PLANE, "icp": _capi.ICP, "vplane": _capi.VPLANE, "ndt": _capi.NDT}
    for name in ("plane", "icp", "vplane", "ndt"):
        if name == "plane":
            make = lambda **kw: pcr.PlaneICP(max_dist=md, k=int(g2["k"]), **kw)
        elif name == "icp":
            make = lambda **kw: pcr.ICP(max_dist=md, **kw)
        elif name == "vplane":
            make = lambda **kw: pcr.VPlaneICP(voxel_size=vs, max_dist=md, **kw)
        else:
            make = lambda **kw: pcr.NDT(voxel_size=vs, max_dist=md, **kw)
        reg = make(devices=devs)
        if name == "plane":
            reg.set_target(target, object(), g2["plane_normals"])     # the reference's own normals (plane_icp.py:25-27)
        else:
            reg.set_target(target)                                     # THE one-line change: devices=[...] above, nothing here
        T = reg.align(source, np.eye(4))
        its = reg.last_iterations
        H, g, e2 = reg.calc_H_g_e2(g2["T"], source)
        assert its == g2[f"align_{name}_T"].shape[0], (name, its)
        assert np.max(np.abs(T[:3, 3] - g2[f"align_{name}_final"][:3, 3])) < 1e-4, name
        Href = g2[f"T_{name}_H"]
        assert np.max(np.abs(H - Href)) < 1e-5 * np.max(np.abs(Href)), name
        # the SPMD sums: every shard on a plain context, local sums, added in rank order from 0.0
        if name in ("plane", "icp"):
            tgt = _capi.Target.points(ctx, target, g2["plane_normals"] if name == "plane" else None)
        else:
            tgt = _capi.Target.voxels(ctx, target, vs, 10)
        flags = _capi.FLAG_ICP_RR_QUIRK | _capi.FLAG_LOCAL_ONLY

        def spmd(Tq):
            tot = np.zeros(29)
            for r in range(n):
                sh = np.ascontiguousarray(pdist.shard_scan(source.astype(np.float32), r, n))
                sc = _capi.Scan(ctx, sh)
                tot = tot + _capi.linearize(tgt, sc, kinds[name], Tq, md, flags)
                sc.close()
            return tot
        out_g = _capi.linearize(reg._target, reg._scan_for(source), kinds[name], g2["T"], md)
        assert np.array_equal(out_g, spmd(g2["T"])), (name, "group sums != rank-ordered sum of the shards' sums")
        # device-resident loop == the C host loop (PCR_FLAG_HOST_LOOP: one pass + gn_step per iteration)
        sc_g = reg._scan_for(source)
        Td, itd = _capi.align(reg._target, sc_g, kinds[name], np.eye(4), reg.max_iter, reg.tol, md, _capi.FLAG_ICP_RR_QUIRK | _capi.FLAG_DEVICE_LOOP)
        Th, ith = _capi.align(reg._target, sc_g, kinds[name], np.eye(4), reg.max_iter, reg.tol, md, _capi.FLAG_ICP_RR_QUIRK | _capi.FLAG_HOST_LOOP)
        assert np.array_equal(Td, Th) and itd == ith == its and np.array_equal(Td, T), (name, "device loop != host loop", itd, ith, its)
        # 70 exchanges in a row: the slot index wraps at 64
        rng = np.random.default_rng(7)
        for i in range(70):
            Tq = np.array(g2["T"], dtype=np.float64)
            Tq[:3, 3] += rng.normal(0, 0.01, 3)
            o = _capi.linearize(reg._target, sc_g, kinds[name], Tq, md)
            if i in (0, 63, 64, 69):
                assert np.array_equal(o, spmd(Tq)), (name, "exchange", i)
        tgt.close()
        print(f"group[{n}] {name}: {its} iterations, H rel err {np.max(np.abs(H - Href)) / np.max(np.abs(Href)):.2e}", flush=True)
    print("GROUP_CHECK_OK", flush=True)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 2)
