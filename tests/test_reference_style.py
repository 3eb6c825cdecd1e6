"""The reference's own four unit tests (tests/test_icp.py, test_picp.py, test_vpicp.py, test_ndt.py),
restated against this package: same fixture (seed 42, 100 random points, R = expSO3([0.1, 0.2, 0.3]),
t = [0.5, -0.3, 0.2]), same assertion -- the vectorised ``calc_H_g_e2`` (here: the HIP kernels) equals
the per-point loop ``calc_H_g_e2_no_parallel_ver`` to atol = 1e-3 at cur_T = I -- plus the cases those
tests cannot see: a multi-voxel masked cloud and the reference's own numbers.
Every test runs once per kernel pipeline (conftest.PIPELINES); "default" is the shipped one."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture
def generate_test_data():
    from point_cloud_registration_amd import expSO3
    np.random.seed(42)
    target = np.random.rand(100, 3)
    R = expSO3(np.array([0.1, 0.2, 0.3]))
    t = np.array([0.5, -0.3, 0.2])
    source = (R @ target.T).T + t
    return target, source


def _compare(obj, source, cur_T=np.eye(4), atol=1e-3):
    H1, g1, e2_1 = obj.calc_H_g_e2(cur_T, source)
    H2, g2, e2_2 = obj.calc_H_g_e2_no_parallel_ver(cur_T, source)
    assert np.allclose(H1, H2, atol=atol), f"H matrices differ: {np.max(np.abs(H1 - H2))}"
    assert np.allclose(g1, g2, atol=atol), f"g vectors differ: {np.max(np.abs(g1 - g2))}"
    assert np.isclose(e2_1, e2_2, atol=atol), f"e2 values differ: {abs(e2_1 - e2_2)}"
    return H1, g1, e2_1


def test_icp_calc_H_g_e2(generate_test_data, g1, pipeline):
    from point_cloud_registration_amd import ICP
    target, source = generate_test_data
    icp = ICP(max_iter=10, max_dist=2.0, tol=1e-3)
    icp.set_target(target)
    H, g, e2 = _compare(icp, source.astype(np.float32))
    assert np.allclose(H, g1["I_icp_H"], atol=1e-3) and abs(e2 - g1["I_icp_e2"]) < 1e-4   # the reference's numbers


def test_plane_icp_calc_H_g_e2(generate_test_data, g1, pipeline):
    from point_cloud_registration_amd import PlaneICP
    target, source = generate_test_data
    picp = PlaneICP(max_iter=10, max_dist=2.0, tol=1e-3)
    picp.set_target(target)
    _compare(picp, source.astype(np.float32))
    picp.set_target(target, picp.kdtree, g1["plane_normals"])        # the reference's normals -> its numbers
    H, g, e2 = _compare(picp, source.astype(np.float32))
    assert np.allclose(H, g1["I_plane_H"], atol=1e-3) and abs(e2 - g1["I_plane_e2"]) < 1e-4


def test_vplane_icp_calc_H_g_e2(generate_test_data, g1, pipeline):
    from point_cloud_registration_amd import VPlaneICP
    target, source = generate_test_data
    vp = VPlaneICP(voxel_size=1.0, max_iter=10, max_dist=2.0, tol=1e-3)
    vp.set_target(target)
    H, g, e2 = _compare(vp, source.astype(np.float32))
    assert np.allclose(H, g1["I_vplane_H"], atol=1e-3) and abs(e2 - g1["I_vplane_e2"]) < 1e-4


def test_ndt_calc_H_g_e2(generate_test_data, g1, pipeline):
    from point_cloud_registration_amd import NDT
    target, source = generate_test_data
    ndt = NDT(voxel_size=1.0, max_iter=10, max_dist=2.0, tol=1e-3)
    ndt.set_target(target)
    H, g, e2 = _compare(ndt, source.astype(np.float32), atol=1e-2)     # |H| ~ 1e3-1e4 here
    assert np.allclose(H, g1["I_ndt_H"], rtol=1e-6, atol=1e-2) and abs(e2 - g1["I_ndt_e2"]) < 1e-3


@pytest.mark.parametrize("name", ["plane", "vplane", "ndt"])
def test_loop_equals_kernels_on_masked_multivoxel_cloud(g2, name, pipeline):
    """What the reference's tests cannot exercise: several voxels, ~10 % of the scan gated out, R != I."""
    import point_cloud_registration_amd as pcr
    md, vs = float(g2["max_dist"]), float(g2["voxel_size"])
    obj = {"plane": pcr.PlaneICP(max_dist=md, k=int(g2["k"])), "vplane": pcr.VPlaneICP(voxel_size=vs, max_dist=md),
           "ndt": pcr.NDT(voxel_size=vs, max_dist=md)}[name]
    obj.set_target(g2["target"])
    H1, g1_, e1 = obj.calc_H_g_e2(g2["T"], g2["source"])
    H2, g2_, e2 = obj.calc_H_g_e2_no_parallel_ver(g2["T"], g2["source"])
    # the loop transforms with NumPy's sgemm, the kernels in a fixed float32 order: ~1e-7 relative
    assert np.max(np.abs(H1 - H2)) < 1e-6 * np.max(np.abs(H1))
    assert np.max(np.abs(g1_ - g2_)) < 1e-5 * max(np.max(np.abs(g1_)), 1.0) and abs(e1 - e2) < 1e-6 * abs(e1)
