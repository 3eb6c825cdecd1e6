import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


@pytest.fixture(scope="session")
def g1():
    return load_golden("g1_reference_fixture.npz")


@pytest.fixture(scope="session")
def g2():
    return load_golden("g2_mini_street.npz")


@pytest.fixture(scope="session")
def g3():
    return load_golden("g3_voxels.npz")


@pytest.fixture(scope="session")
def g5():
    return load_golden("g5_se3.npz")


@pytest.fixture(scope="session")
def g6():
    return load_golden("g6_normals.npz")


@pytest.fixture(scope="session")
def g7():
    """Full-scale normals fixture + the cloud it refers to (regenerated, checksum-guarded)."""
    import zlib
    from point_cloud_registration_amd.synthetic import street
    g = load_golden("g7_normals_fullscale.npz")
    pts = street(int(g["n"]), seed=0)
    assert zlib.crc32(pts.tobytes()) == int(g["crc32"]), "street() no longer reproduces the cloud of the fixture"
    g["points"] = pts
    return g


@pytest.fixture(scope="session")
def g9():
    return load_golden("g9_q6_f64_target.npz")


@pytest.fixture(scope="session")
def g8():
    """BASELINE-size fixture (reference run on the 1.06 M-point B-01 stand-in) + the clouds it refers to,
    regenerated from the deterministic generators and checksum-guarded."""
    import zlib
    from point_cloud_registration_amd.synthetic import street, harness_scan, perturbed_scan, street_normals
    g = load_golden("g8_b01_fullsize.npz")
    target = street(int(g["n"]), seed=0)
    clouds = {"target": target, "harness100k": harness_scan(target, 100_000, seed=1),
              "pert100k": perturbed_scan(target, 100_000, seed=2)[0],
              "pertfull": perturbed_scan(target, None, seed=2)[0], "given_normals": street_normals(target)}
    for name, arr in clouds.items():
        assert zlib.crc32(arr.tobytes()) == int(g[f"crc32_{name}"]), f"{name}: the generator no longer reproduces the fixture's cloud"
    g.update(clouds)
    return g


@pytest.fixture(scope="session")
def g10():
    """BASELINE configs[2] / [3] at config size (reference run on the 10 M-point cloud and its FULL 10 M-point perturbed
    scan -- bench.py's vplane_10m / ndt_10m workloads) + the clouds, regenerated and checksum-guarded."""
    import zlib
    from point_cloud_registration_amd.synthetic import street_tiled, perturbed_scan
    g = load_golden("g10_10m_voxel.npz")
    target = street_tiled(int(g["n"]), seed=0)
    scan = perturbed_scan(target, None, seed=2)[0]
    assert zlib.crc32(target.tobytes()) == int(g["crc32_target"]), "street_tiled() no longer reproduces the fixture's cloud"
    assert zlib.crc32(scan.tobytes()) == int(g["crc32_scan"]), "perturbed_scan() no longer reproduces the fixture's scan"
    g["target"], g["scan"] = target, scan
    return g


@pytest.fixture(scope="session")
def g11():
    """Non-uniform density: the reference's four classes on a 200 k-point LiDAR sweep (synthetic.lidar_sweep) + the clouds,
    regenerated and checksum-guarded."""
    import zlib
    from point_cloud_registration_amd.synthetic import lidar_sweep, lidar_normals, perturbed_scan
    g = load_golden("g11_lidar_sweep.npz")
    target = lidar_sweep(int(g["n"]), seed=0)
    scan = perturbed_scan(target, int(g["n_scan"]), seed=2)[0]
    normals = lidar_normals(target)
    assert zlib.crc32(target.tobytes()) == int(g["crc32_target"]), "lidar_sweep() no longer reproduces the fixture's cloud"
    assert zlib.crc32(scan.tobytes()) == int(g["crc32_scan"]) and zlib.crc32(normals.tobytes()) == int(g["crc32_normals"])
    g["target"], g["scan"], g["given_normals"] = target, scan, normals
    return g


@pytest.fixture(scope="session")
def g12():
    return load_golden("g12_voxel_filter.npz")


def step_err(H, g, Href, gref):
    """What a difference in (H, g) does to the Gauss-Newton step of registration.py:103: |solve(H, g) - solve(Href, gref)|_inf
    (metres / radians).  Unlike max|dg| / max|g_k| it stays meaningful at the converged poses, where g itself cancels to
    rounding level."""
    return float(np.max(np.abs(np.linalg.solve(H, g) - np.linalg.solve(Href, gref))))


def rel_H(H, Href):
    """Parity metric of SURVEY.md section 8a Q5: max|dH| / max|H_ref|."""
    return float(np.max(np.abs(np.asarray(H) - np.asarray(Href))) / np.max(np.abs(Href)))


# Kernel pipelines of the hot path (pcr_set_variant / pcr_set_fuse_finalize / pcr_set_nn_mode).  "default" is
# what ships and what bench.py times; the others are kept selectable for A/B measurements and must
# produce the same sums.
PIPELINES = {
    "default": dict(variant=2, fuse_finalize=1, nn_mode=0, reuse=0),    # per launch: fused kernel for small scans, search + reduce for large
    "split": dict(variant=1, fuse_finalize=1, nn_mode=0, reuse=0),      # k_nn_scan + k_reduce_finalize (what large scans run)
    "reuse_auto": dict(variant=1, fuse_finalize=1, nn_mode=0, reuse=1), # ... with certified reuse under its automatic policy (opt-in since round 5)
    "reuse": dict(variant=1, fuse_finalize=1, nn_mode=0, reuse=2),      # ... certified reuse of the previous matches FORCED on every pass it can run on
    "noreuse": dict(variant=1, fuse_finalize=1, nn_mode=0, reuse=0),    # ... and off
    "mfma": dict(variant=1, fuse_finalize=1, nn_mode=4, reuse=0),       # wave-cooperative search with an MFMA distance filter (k_nn_mfma, round 5)
    "coop": dict(variant=1, fuse_finalize=1, nn_mode=2, reuse=0),       # wave-cooperative search (k_nn_coop)
    "nofilter": dict(variant=1, fuse_finalize=1, nn_mode=3, reuse=0),   # centroid searches in float64 throughout (no float32 filter + check)
    "unfused": dict(variant=1, fuse_finalize=0, nn_mode=0, reuse=2),    # k_nn_scan + k_reduce + k_finalize
    "onekernel": dict(variant=0, fuse_finalize=1, nn_mode=0, reuse=0),  # k_linearize_finalize (what small scans run)
    "onekernel_unfused": dict(variant=0, fuse_finalize=0, nn_mode=0, reuse=0),   # k_linearize + k_finalize
}


# pipelines that need the developer kernels of csrc/kernels_dev.hip: only libpcr_hip_dev.so (`make dev`) has them.  The
# shipped library runs the others; tests/test_gpu_dev_build.py re-runs every pipeline test in a process that loaded the
# developer build (PCR_LIB).
DEV_PIPELINES = ("coop", "mfma", "unfused", "onekernel_unfused")


@pytest.fixture(params=list(PIPELINES))
def pipeline(request):
    """Runs the test once per kernel pipeline on the process-wide context of device 0, restoring the
    shipped selection afterwards."""
    from point_cloud_registration_amd import _capi
    if request.param in DEV_PIPELINES and not _capi.has_dev_kernels():
        pytest.skip("developer pipeline: runs in the developer build (tests/test_gpu_dev_build.py)")
    ctx = _capi.get_context(0)
    before = ctx.get_pipeline()
    with ctx.pipeline(**PIPELINES[request.param]):
        yield request.param
    assert ctx.get_pipeline() == before
