"""Pin the CPU oracle (oracle/) against golden vectors produced by the reference itself
(tests/golden/make_golden.py).  No GPU needed."""

import os

import numpy as np
import pytest

from conftest import step_err, rel_H, REPO
from oracle import oracle as orc

KINDS = {"icp": orc.ICP, "plane": orc.PLANE, "vplane": orc.VPLANE, "ndt": orc.NDT}
# stated tolerance (BASELINE.json north_star): J^T J within 1e-5 relative (max|dH|/max|H|)
TOL_H = 1e-5


def _targets(g, with_ref_normals=True):
    pts = orc.TargetPoints(g["target"], normals=g["plane_normals"] if with_ref_normals else None)
    vox = orc.TargetVoxels(g["target"], float(g["voxel_size"]))
    return {"icp": pts, "plane": pts, "vplane": vox, "ndt": vox}


@pytest.mark.parametrize("name", list(KINDS))
@pytest.mark.parametrize("tag", ["I", "T"])
def test_g1_reference_fixture(g1, name, tag):
    """The reference tests' own fixture (tests/test_icp.py:7-17 ...), at identity and at R != I
    (the latter exposes quirk Q1, which the reference's own tests cannot see)."""
    tg = _targets(g1)[name]
    T = np.eye(4) if tag == "I" else g1["T"]
    H, g, e2 = orc.calc_H_g_e2(KINDS[name], tg, T, g1["source"], float(g1["max_dist"]))
    assert rel_H(H, g1[f"{tag}_{name}_H"]) < TOL_H
    assert rel_H(g, g1[f"{tag}_{name}_g"]) < TOL_H
    assert abs(e2 - g1[f"{tag}_{name}_e2"]) < TOL_H * max(1.0, abs(g1[f"{tag}_{name}_e2"]))


def test_g1_known_answers(g1):
    """Known-answer anchors recorded in SURVEY.md section 4."""
    tg = _targets(g1)
    H, g, e2 = orc.calc_H_g_e2(orc.ICP, tg["icp"], np.eye(4), g1["source"], 2.0)
    assert np.allclose(np.diag(H), [100, 100, 100, 70.998062, 142.096771, 108.251877], atol=2e-4)
    assert np.allclose(g, [13.442728, -7.253846, 3.631026, 6.271748, 5.654261, -10.432824], atol=2e-5)
    assert abs(e2 - 7.569768) < 1e-5
    assert abs(orc.calc_H_g_e2(orc.VPLANE, tg["vplane"], np.eye(4), g1["source"], 2.0)[2] - 28.485575) < 1e-5
    Hn, _, e2n = orc.calc_H_g_e2(orc.NDT, tg["ndt"], np.eye(4), g1["source"], 2.0)
    assert abs(e2n - 632.083526) < 1e-4 and abs(Hn[0, 0] - 1236.125756) < 1e-4
    assert np.allclose(tg["vplane"].mean[0], [0.476849, 0.512461, 0.496303], atol=1e-6)


def test_g1_icp_quirk_q1(g1):
    """At R != I the vectorised reference gradient is sum p x (R r); the consistent J^T r
    (its loop version) differs -- both are available, the quirk is the default."""
    tg = _targets(g1)["icp"]
    T = g1["T"]
    _, gq, _ = orc.calc_H_g_e2(orc.ICP, tg, T, g1["source"], 2.0, flags=orc.FLAG_ICP_RR_QUIRK)
    _, gc, _ = orc.calc_H_g_e2(orc.ICP, tg, T, g1["source"], 2.0, flags=0)
    assert rel_H(gq, g1["T_icp_g"]) < TOL_H
    assert np.max(np.abs(gq[3:] - gc[3:])) > 1e-3          # they really differ
    assert np.allclose(gq[:3], gc[:3])
    # at identity the loop oracle of the reference is valid and agrees with the consistent form
    _, gI, _ = orc.calc_H_g_e2(orc.ICP, tg, np.eye(4), g1["source"], 2.0, flags=0)
    assert rel_H(gI, g1["I_icp_loop_g"]) < TOL_H


@pytest.mark.parametrize("name", list(KINDS))
def test_g2_masked_multivoxel(g2, name):
    """5 k-point mini street, 2 k scan with outliers, R != I, ~9-11 % of points gated out."""
    tg = _targets(g2)[name]
    H, g, e2 = orc.calc_H_g_e2(KINDS[name], tg, g2["T"], g2["source"], float(g2["max_dist"]))
    assert rel_H(H, g2[f"T_{name}_H"]) < TOL_H
    assert rel_H(g, g2[f"T_{name}_g"]) < 5 * TOL_H
    assert abs(e2 - g2[f"T_{name}_e2"]) < 5 * TOL_H * abs(g2[f"T_{name}_e2"])


def test_g2_correspondences(g2):
    """Exact NN against the reference's KD-tree answers (ties compared by distance)."""
    st = orc.transform(g2["T"], g2["source"])
    d, i = orc.nn_brute(g2["target"], st)
    same = i == g2["nn_idx"]
    assert same.mean() > 0.999
    # reference transform = BLAS sgemm, ours = fixed-order float32: coordinates differ by ~1 ulp
    assert np.allclose(d, g2["nn_dist"], rtol=1e-5, atol=2e-6)
    vox = orc.TargetVoxels(g2["target"], float(g2["voxel_size"]))
    dv, iv = vox.query(st)
    assert (iv == g2["vox_idx"]).mean() > 0.999
    assert np.allclose(dv, g2["vox_dist"], rtol=1e-5, atol=2e-6)
    # grid-accelerated search == exhaustive search, bit for bit, bounded and unbounded
    grid = orc.Grid(g2["target"], 0.2)
    dg, ig = grid.query(st)
    assert np.array_equal(ig, i) and np.array_equal(dg.astype(np.float32), d)
    dg, ig = grid.query(st, r_max=0.8)
    keep = d < 0.8
    assert np.array_equal(ig[keep], i[keep]) and np.all(ig[~keep] == -1)
    gv = orc.Grid(vox.mean, 1.0)
    dgv, igv = gv.query(st)
    assert np.array_equal(igv, iv) and np.array_equal(dgv, dv)


@pytest.mark.parametrize("name", list(KINDS))
def test_g2_align_trajectory(g2, name):
    """Full Gauss-Newton runs: same iteration count, per-iteration H/g/e2 and final SE(3)
    within the north-star tolerance (1e-4 rad / 1e-4 m)."""
    tg = _targets(g2)[name]
    trace = []
    T = orc.align(KINDS[name], tg, g2["source"], np.eye(4), max_iter=30, tol=1e-3,
                  max_dist=float(g2["max_dist"]), trace=trace)
    ref_T = g2[f"align_{name}_T"]
    assert len(trace) == ref_T.shape[0]
    for it, (cur, H, g, e2) in enumerate(trace):
        assert np.allclose(cur, ref_T[it], atol=1e-5)
        assert rel_H(H, g2[f"align_{name}_H"][it]) < 1e-4     # trajectories drift by rounding
    final = g2[f"align_{name}_final"]
    assert np.max(np.abs(T[:3, 3] - final[:3, 3])) < 1e-4
    dR = T[:3, :3] @ final[:3, :3].T
    ang = np.arccos(np.clip((np.trace(dR) - 1) / 2, -1, 1))
    assert ang < 1e-4


@pytest.mark.parametrize("dt", ["f32", "f64"])
@pytest.mark.parametrize("vs", [0.5, 1.0])
def test_g3_voxel_build(g3, dt, vs):
    tag = f"{dt}_vs{vs}"
    pts = g3[f"points_{dt}"]
    keys = orc.voxel_keys(pts, vs)
    assert np.array_equal(keys, g3[f"{tag}_keys"])                       # integer work: bit-exact
    v = orc.voxel_build(pts, vs, 10)
    assert v["n_unique"] == len(g3[f"{tag}_uniq"])
    counts = g3[f"{tag}_counts"]
    assert np.array_equal(v["counts"], counts[counts >= 10])
    assert np.array_equal(v["keys"], g3[f"{tag}_uniq"][counts >= 10])
    assert np.allclose(v["mean"], g3[f"{tag}_mean"], rtol=0, atol=1e-12)
    assert np.allclose(v["cov"], g3[f"{tag}_cov"], rtol=1e-10, atol=1e-15)
    icov = orc.calc_icov(v["cov"])
    assert np.allclose(icov, g3[f"{tag}_icov"], rtol=1e-7, atol=0)
    # normals: sign-free, only where the smallest eigenvalue is well separated
    ev = g3[f"{tag}_evals"]
    ok = (ev[:, 1] - ev[:, 0]) > 1e-3 * ev[:, 2]
    dots = np.abs(np.sum(v["norm"] * g3[f"{tag}_norm"], axis=1))
    assert ok.sum() > 0.8 * len(ok)
    assert np.all(dots[ok] > 1 - 1e-8)


def test_g5_se3(g5):
    for w, R in zip(g5["omegas"], g5["Rs"]):
        assert np.allclose(orc.expSO3(w), R, atol=1e-15)
    for dx, T in zip(g5["dxs"], g5["Ts"]):
        assert np.allclose(orc.plus(g5["T0"], dx), T, atol=1e-15)
    # host mirror used by the product
    from point_cloud_registration_amd import math_tools as mt
    for w, R in zip(g5["omegas"], g5["Rs"]):
        assert np.allclose(mt.expSO3(w), R, atol=1e-15)
    for dx, T in zip(g5["dxs"], g5["Ts"]):
        assert np.allclose(mt.plus(g5["T0"], dx), T, atol=1e-15)


def test_solve6_and_singular():
    rng = np.random.default_rng(0)
    A = rng.normal(size=(6, 6)); H = A @ A.T + np.eye(6); g = rng.normal(size=6)
    assert np.allclose(orc.solve6(H, g), np.linalg.solve(H, g), rtol=1e-12)
    with pytest.raises(np.linalg.LinAlgError):
        orc.solve6(np.zeros((6, 6)), g)                                    # quirk Q7


@pytest.mark.parametrize("k", [5, 15])
def test_g6_normals(g6, k):
    """k-NN PCA normals in the reference's float32 single-pass arithmetic (compat)."""
    pts = g6["points"]
    _, idx = orc.knn_brute(pts, pts, k)
    n = orc.normals_from_knn(pts, idx, compat=True)
    dots = np.abs(np.sum(n * g6[f"normals_k{k}"], axis=1))
    # (the thresholds of the full-scale g7 test; measured here: every normal within 1e-7 of the reference's)
    assert np.mean(dots > 0.999) >= 0.999, np.mean(dots > 0.999)
    assert np.mean(dots > 1 - 1e-5) > 0.99


@pytest.mark.parametrize("k", [5, 15])
def test_g7_normals_full_scale(g7, k):
    """The float32 single-pass covariance at B-01 scale (|p| up to 67 m, real density): the oracle's
    compat normals are the reference's on (almost) every sampled point, far corners included."""
    pts = g7["points"]
    far = np.argsort(-np.linalg.norm(pts[g7["sample"]], axis=1))[:500]
    pick = np.unique(np.concatenate([np.arange(1000), far]))
    idx_pts = g7["sample"][pick]
    _, idx = orc.knn_brute(pts, pts[idx_pts], k)
    n = orc.normals_from_knn(pts, idx, compat=True)
    dots = np.abs(np.sum(n.astype(np.float64) * g7[f"normals_k{k}"][pick], axis=1))
    assert np.mean(dots > 0.999) >= 0.999, np.mean(dots > 0.999)
    assert np.mean(dots > 1 - 1e-5) > 0.99


# ----------------------------------------------------------------------------- g8: BASELINE size
def _pose_err(T, ref):
    dR = T[:3, :3] @ ref[:3, :3].T
    ang = np.arccos(np.clip((np.trace(dR) - 1) / 2, -1, 1))
    return float(np.max(np.abs(T[:3, 3] - ref[:3, 3]))), float(ang)


@pytest.mark.parametrize("scan_name", ["harness100k", "pert100k"])
def test_g8_oracle_matches_reference_at_b01_size(g8, scan_name):
    """the 1e-5 / 1e-4 bars against the REFERENCE at the size the metric is quoted on (1.06 M-point
    target, |p| up to 67 m, M = 1e5 correspondences, where the reference's float32 partial sums are largest): the
    oracle's H, g, e2 at the identity and at every mid pose of the reference's own align() trajectory, then the
    oracle's own align(): same iteration count, same final pose.  (PlaneICP with the supplied analytic normals: the
    oracle's brute-force k-NN cannot estimate 1.06 M normals in a test; the reference's own-normal variant is
    checked on the GPU.)"""
    target, scan = g8["target"], g8[scan_name]
    md = float(g8["max_dist"])
    tp = orc.TargetPoints(target, normals=g8["given_normals"], cell=0.5)
    tv = orc.TargetVoxels(target, float(g8["voxel_size"]))
    assert tv.mean.shape[0] == int(g8["n_voxels"])
    for cname, kind, tgt in (("icp", orc.ICP, tp), ("planeg", orc.PLANE, tp), ("vplane", orc.VPLANE, tv), ("ndt", orc.NDT, tv)):
        tag = f"{scan_name}_{cname}"
        Ts = g8[f"{tag}_T"]
        for k in range(Ts.shape[0]):
            H, g, e2 = orc.calc_H_g_e2(kind, tgt, Ts[k], scan, md)
            assert rel_H(H, g8[f"{tag}_H"][k]) <= 1e-5, (tag, k, rel_H(H, g8[f"{tag}_H"][k]))
            # g cancels towards convergence: relative to the gradient at the start of the run ...
            assert np.max(np.abs(g - g8[f"{tag}_g"][k])) <= 1e-4 * np.max(np.abs(g8[f"{tag}_g"][0])), (tag, k)
            # ... and, at EVERY pose, by what it does to the step the reference takes from here
            assert step_err(H, g, g8[f"{tag}_H"][k], g8[f"{tag}_g"][k]) <= 5e-5, (tag, k)
            assert abs(e2 - g8[f"{tag}_e2"][k]) <= 1e-4 * abs(g8[f"{tag}_e2"][k]), (tag, k)
        trace = []
        T = orc.align(kind, tgt, scan, max_iter=30, tol=1e-3, max_dist=md, trace=trace)
        assert len(trace) == Ts.shape[0], (tag, len(trace), Ts.shape[0])
        dt, dr = _pose_err(T, g8[f"{tag}_final"])
        assert dt <= 1e-4 and dr <= 1e-4, (tag, dt, dr)


def test_g11_oracle_matches_reference_on_lidar_sweep(g11):
    """Non-uniform density: the reference's four classes on one LiDAR revolution (density ~ 1/r^2, ring
    lines; 200 k-point map, 50 k-point scan).  H <= 1e-5, the Gauss-Newton step <= 5e-5 at every iterate of the reference's own
    align() -- including NDT's 30 iterations, which do NOT converge on this cloud in the reference either (singular voxel
    covariances on line-shaped voxels, ndt.py:24-57 has no regularisation): parity means following it there."""
    target, scan, md = g11["target"], g11["scan"], float(g11["max_dist"])
    tp = orc.TargetPoints(target, normals=g11["given_normals"], cell=0.5)
    tv = orc.TargetVoxels(target, float(g11["voxel_size"]))
    assert tv.mean.shape[0] == int(g11["n_voxels"])
    for cname, kind, tgt in (("icp", orc.ICP, tp), ("planeg", orc.PLANE, tp), ("vplane", orc.VPLANE, tv), ("ndt", orc.NDT, tv)):
        Ts = g11[f"{cname}_T"]
        for k in range(Ts.shape[0]):
            H, g, e2 = orc.calc_H_g_e2(kind, tgt, Ts[k], scan, md)
            assert rel_H(H, g11[f"{cname}_H"][k]) <= 1e-5, (cname, k)
            assert np.max(np.abs(g - g11[f"{cname}_g"][k])) <= 1e-4 * np.max(np.abs(g11[f"{cname}_g"][0])), (cname, k)
            assert abs(e2 - g11[f"{cname}_e2"][k]) <= 1e-4 * abs(g11[f"{cname}_e2"][k]), (cname, k)
            assert step_err(H, g, g11[f"{cname}_H"][k], g11[f"{cname}_g"][k]) <= 5e-5, (cname, k)
        trace = []
        T = orc.align(kind, tgt, scan, max_iter=30, tol=1e-3, max_dist=md, trace=trace)
        assert len(trace) == Ts.shape[0], (cname, len(trace), Ts.shape[0])
        dt, dr = _pose_err(T, g11[f"{cname}_final"])
        assert dt <= 1e-4 and dr <= 1e-4, (cname, dt, dr)


@pytest.mark.parametrize("cname,vs", [("vplane", 0.5), ("ndt", 1.0)])
def test_g10_oracle_matches_reference_at_10m(g10, cname, vs):
    """VERDICT r4 missing #4: BASELINE configs[2] (VPlaneICP, voxel 0.5) and configs[3] (NDT, voxel 1.0) AT CONFIG SIZE --
    the reference itself (voxelized_plane_icp.py:23-64, ndt.py:24-57, voxel.py:104-179) ran on the 10 M-point cloud and the
    full 10 M-point scan; the oracle's voxel build keeps the same number of voxels with the same means, and its H, g, e2
    over all 10 M scan points meet the 1e-5 / 1e-4 bars at the identity, a mid pose and T_true (~10 s per class here)."""
    kind = {"vplane": orc.VPLANE, "ndt": orc.NDT}[cname]
    tv = orc.TargetVoxels(g10["target"], vs)
    assert tv.mean.shape[0] == int(g10[f"{cname}_n_voxels"])
    assert np.allclose(tv.mean[::997], g10[f"{cname}_mean_sample"], rtol=0, atol=1e-9)
    for k, T in enumerate(g10["poses"]):
        H, g, e2 = orc.calc_H_g_e2(kind, tv, T, g10["scan"], float(g10["max_dist"]))
        assert rel_H(H, g10[f"{cname}_H"][k]) <= 1e-5, (cname, k, rel_H(H, g10[f"{cname}_H"][k]))
        assert np.max(np.abs(g - g10[f"{cname}_g"][k])) <= 1e-4 * np.max(np.abs(g10[f"{cname}_g"][0])), (cname, k)
        assert abs(e2 - g10[f"{cname}_e2"][k]) <= 1e-4 * abs(g10[f"{cname}_e2"][k]), (cname, k)


def test_g13_fixture_and_tiled_normals():
    """g13 = the reference itself on BASELINE configs[4] at config size (1e8-point target, 12.5 M-point shard; make_golden.py: g13).
    The 1e8-point runs live elsewhere -- the HIP path in tests/test_gpu_fullsize.py::test_100m_plane (GPU), the oracle in
    tools/g13_oracle_check.py (10 min, 10 GB: profiles/r06_g13_parity.txt; PCR_SLOW=1 runs it from here).  This test pins what
    both regenerate: the fixture's shape, and the tile-wise analytic normals against the single-street ones."""
    from conftest import load_golden
    from point_cloud_registration_amd.synthetic import street, street_normals, street_tiled, street_tiled_normals
    g = load_golden("g13_100m_plane.npz")
    assert int(g["n"]) == 100_000_000 and int(g["n_scan"]) == 12_500_000 and g["poses"].shape == (3, 4, 4)
    for c in ("planeg", "icp"):
        assert g[f"{c}_H"].shape == (3, 6, 6) and g[f"{c}_g"].shape == (3, 6) and g[f"{c}_e2"].shape == (3,)
        assert np.all(np.linalg.eigvalsh(g[f"{c}_H"][2]) > 0) and g[f"{c}_e2"][2] < g[f"{c}_e2"][0]
    assert g["icp_H"][2][0, 0] == 12_500_000                      # every scan point matched at T_true
    t = street_tiled(2_000_000, seed=0)
    n = street_tiled_normals(t)
    assert n.dtype == np.float32 and set(np.unique(n)) == {0.0, 1.0} and np.all(n.sum(1) == 1.0)
    one = street(1_000_000, seed=0 * 100003 + 0, center=(0.0, 0.0))
    assert np.mean(np.all(street_normals(one) == n[:1_000_000], axis=1)) > 0.9999      # (corner points within 0.2 m of two walls may differ)
    if os.environ.get("PCR_SLOW") == "1":
        import subprocess, sys
        r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "g13_oracle_check.py")], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-800:]
        print(r.stdout)


def test_g9_quirk_q6_float64_target(g9):
    """Quirk Q6, reproduced since round 5.  The reference's PlaneICP searches a tree built on the ORIGINAL float64 array
    (plane_icp.py:22) and gathers from the float32 copy (plane_icp.py:20,44).  The fixture makes that bite: ~720 queries
    within ~2e-5 m of the bisector plane of two target points whose coordinates (~500 m) are not float32-representable --
    82 of them get a different neighbour under the float64 tree, and H moves by 2.5e-3.  Pinned here:
    (i) with tree_f64 the oracle returns the REFERENCE's float64-tree neighbour for every query, near-ties included, and
        H, g, e2 of the reference's own class within the usual bars at all three poses;
    (ii) on the float32 copy (what ICP.set_target searches, icp.py:19-20) it returns the reference's float32-tree
        neighbour for every query and the reference class evaluated on that tree;
    (iii) the two differ where the fixture says they do (the fixture CAN fail for the reason it exists)."""
    i64, i32 = g9["nn_idx_f64_tree"], g9["nn_idx_f32_tree"]
    n_ord = int(g9["n_ordinary"])
    differ = i64 != i32
    assert differ.sum() >= 50 and not differ[:n_ord].any()            # (iii)
    t32 = g9["target"].astype(np.float32)
    md = float(g9["max_dist"])
    I = np.eye(4)
    st = orc.transform(I, g9["source_tie"])
    assert np.array_equal(st, g9["source_tie"])                        # the float32 transform is exact at the identity
    # (i) the float64 tree
    tq = orc.TargetPoints(g9["target"], normals=g9["plane_normals"], tree_f64=True)
    d, i = tq.query(st)
    db, ib = orc.nn_brute_f64(g9["target"], st)
    assert np.array_equal(i, ib) and np.array_equal(d, db)
    assert np.array_equal(i, i64)
    assert np.allclose(d, g9["nn_dist_f64_tree"], rtol=1e-12)
    for tag, T, sc in (("T", g9["T"], "source"), ("N", g9["T_near"], "source"), ("E", I, "source_tie")):
        H, g, e2 = orc.calc_H_g_e2(orc.PLANE, tq, T, g9[sc], md)
        assert rel_H(H, g9[f"{tag}_plane_H"]) < TOL_H, tag
        assert np.max(np.abs(g - g9[f"{tag}_plane_g"])) < 10 * TOL_H * np.max(np.abs(g9["N_plane_g"]))
        # e2 at poses T / N: the float32 transform itself rounds to 3e-5 m at these coordinates, in a summation order
        # that differs between NumPy's matmul and the build's (A2) -- 1.5 % of a 2 mm residual per point
        assert abs(e2 - g9[f"{tag}_plane_e2"]) < (5 * TOL_H if tag == "E" else 2e-3) * abs(g9[f"{tag}_plane_e2"])
    # (ii) the float32 tree
    tp = orc.TargetPoints(t32, normals=g9["plane_normals"])
    d, i = tp.query(st)
    db, ib = orc.nn_brute(t32, st)
    assert np.array_equal(i, ib)                                       # the oracle's grid search = exhaustive search
    assert np.array_equal(i, i32)
    assert np.allclose(d, g9["nn_dist_f32_tree"], rtol=1e-6)
    for tag, T, sc in (("T", g9["T"], "source"), ("N", g9["T_near"], "source"), ("E", I, "source_tie")):
        H, g, e2 = orc.calc_H_g_e2(orc.PLANE, tp, T, g9[sc], md)
        assert rel_H(H, g9[f"{tag}_plane_H_f32tree"]) < TOL_H
        assert np.max(np.abs(g - g9[f"{tag}_plane_g_f32tree"])) < 10 * TOL_H * np.max(np.abs(g9["N_plane_g"]))
        assert abs(e2 - g9[f"{tag}_plane_e2_f32tree"]) < (5 * TOL_H if tag == "E" else 2e-3) * abs(g9[f"{tag}_plane_e2_f32tree"])
    H, g, e2 = orc.calc_H_g_e2(orc.PLANE, tp, I, g9["source_tie"], md)
    assert 1e-4 < rel_H(H, g9["E_plane_H"]) < 1e-2                     # (iii) what ignoring Q6 would cost on this fixture
    T_fin = orc.align(orc.PLANE, tq, g9["source"], g9["T_near"], 30, 1e-3, md)
    # (compared where the data is: 500 m from the origin a rotation difference of 3e-7 rad moves the translation column
    # by 1.5e-4 m although no scan point moves by more than a few 1e-5 m)
    src = g9["source"].astype(np.float64)
    moved = np.linalg.norm((src @ T_fin[:3, :3].T + T_fin[:3, 3]) - (src @ g9["align_final"][:3, :3].T + g9["align_final"][:3, 3]), axis=1)
    dt, dr = _pose_err(T_fin, g9["align_final"])
    assert moved.max() < 1e-4 and dr < 1e-4, (moved.max(), dt, dr)
