"""The developer / A-B kernels (csrc/kernels_dev.hip: unfused folds, the wave-cooperative LDS-staged search of the north
star, the work counters of the search) are NOT in the shipped libpcr_hip.so; `make dev` links them into
libpcr_hip_dev.so.  This test re-runs every pipeline-parametrised parity test for the developer pipelines in a process
that loaded that build (PCR_LIB), so that every alternative DESIGN.md measures against stays runnable and exact."""

import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import REPO, DEV_PIPELINES

pytestmark = pytest.mark.gpu


def test_shipped_library_has_no_developer_kernels():
    from point_cloud_registration_amd import _capi
    if os.environ.get("PCR_LIB"):
        pytest.skip("PCR_LIB selects another build")
    assert not _capi.has_dev_kernels()
    ctx = _capi.get_context(0)
    before = ctx.get_pipeline()
    with pytest.raises(ValueError):
        ctx.set_nn_mode(2)                       # wave-cooperative search: developer build only
    with pytest.raises(ValueError):
        ctx.set_nn_mode(4)                       # ... and its MFMA-filtered relative (round 5)
    with pytest.raises(ValueError):
        ctx.set_fuse_finalize(0)                 # unfused folds: developer build only
    assert ctx.get_pipeline() == before
    out = subprocess.run(["nm", "-D", "--defined-only", _capi.LIB_PATH], check=True, capture_output=True, text=True).stdout
    assert "k_nn_coop" not in out and "pcr_dev_" not in out
    data = open(_capi.LIB_PATH, "rb").read()
    for name in (b"k_nn_coop", b"k_nn_mfma", b"k_nn_bound", b"k_nn_counters", b"10k_finalize", b"8k_nn_fix"):
        assert name not in data, name          # not even as device code in the fat binary


def test_developer_pipelines_in_the_developer_build():
    from point_cloud_registration_amd import _capi
    assert os.path.exists(_capi.DEV_LIB_PATH), "make dev (csrc/Makefile) builds libpcr_hip_dev.so"
    env = dict(os.environ, PCR_LIB=_capi.DEV_LIB_PATH)
    expr = " or ".join(DEV_PIPELINES) + " or nn_counters or fuzz_against_oracle"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(REPO, "tests", "test_gpu_parity.py"),
                        os.path.join(REPO, "tests", "test_reference_style.py"), os.path.join(REPO, "tests", "test_gpu_dev_build.py"),
                        "-m", "gpu", "-x", "-q", "-k", expr, "-p", "no:cacheprovider"],
                       capture_output=True, text=True, timeout=1800, env=env, cwd=REPO)
    tail = r.stdout[-1500:]
    assert r.returncode == 0, tail + r.stderr[-500:]
    assert " passed" in tail and "skipped" not in tail.splitlines()[-1], tail      # they ran, none was skipped
    print(tail.splitlines()[-1])


def test_nn_counters_of_the_developer_build():
    """pcr_nn_counters (work counters of the search: rings, rows, candidates per query) -- developer build only."""
    from point_cloud_registration_amd import _capi
    from point_cloud_registration_amd.synthetic import street, perturbed_scan
    if not _capi.has_dev_kernels():
        ctx = _capi.get_context(0)
        tgt = _capi.Target.points(ctx, street(20000, seed=1))
        sc = _capi.Scan(ctx, perturbed_scan(street(20000, seed=1), 5000, seed=2)[0])
        with pytest.raises(ValueError):
            _capi.nn_counters(tgt, sc, np.eye(4), 2.0)
        return
    ctx = _capi.get_context(0)
    target = street(50000, seed=1)
    tgt = _capi.Target.points(ctx, target)
    sc = _capi.Scan(ctx, perturbed_scan(target, 20000, seed=2)[0])
    c = _capi.nn_counters(tgt, sc, np.eye(4), 2.0)
    assert c["candidates"] > 0 and c["rings"] >= 1.0          # per-query averages
