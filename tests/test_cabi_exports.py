"""The C-ABI library builds for gfx950, loads without a GPU, exports every symbol include/pcr.h
declares, and fails LOUDLY (no CPU fallback) when no MI355X is present."""

import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import REPO


@pytest.fixture(scope="module")
def capi():
    from point_cloud_registration_amd import _capi
    if not os.path.exists(_capi.LIB_PATH):
        from point_cloud_registration_amd.build import build
        build()
    return _capi


def declared_symbols():
    text = open(os.path.join(REPO, "include", "pcr.h")).read()
    return sorted(set(re.findall(r"PCR_API\s+[\w\s\*]+?\b(pcr_\w+)\s*\(", text)))


def test_header_symbols_are_exported(capi):
    names = declared_symbols()
    assert len(names) >= 30
    out = subprocess.run(["nm", "-D", "--defined-only", capi.LIB_PATH], check=True, capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (pcr_\w+)", out))
    missing = [n for n in names if n not in exported]
    assert not missing, f"declared in include/pcr.h but not exported: {missing}"
    # nothing but the declared ABI leaks out of the shared object
    assert exported == set(names), exported ^ set(names)


def test_python_binding_covers_the_header(capi):
    L = capi.lib()
    assert sorted(capi.PROTOTYPES) == declared_symbols()
    for name in capi.PROTOTYPES:
        assert getattr(L, name) is not None
    assert b"gfx950" in L.pcr_version()


def test_library_targets_gfx950(capi):
    """The fat binary inside the .so carries gfx950 code objects and nothing else."""
    data = open(capi.LIB_PATH, "rb").read()
    archs = set(re.findall(rb"amdgcn-amd-amdhsa--(gfx[0-9a-f]+)", data))
    assert archs == {b"gfx950"}, archs


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_fails_loudly_without_a_gpu(capi):
    assert capi.device_count() == 0
    h = ctypes.c_void_p()
    rc = capi.lib().pcr_context_create(0, ctypes.byref(h))
    assert rc != 0 and capi.lib().pcr_last_error()
    import point_cloud_registration_amd as pcr
    pts = np.random.default_rng(0).normal(size=(100, 3)).astype(np.float32)
    for cls in (pcr.ICP, pcr.PlaneICP, pcr.VPlaneICP, pcr.NDT):
        with pytest.raises((capi.PcrError, ValueError)):
            cls().set_target(pts)                      # no silent CPU path
    with pytest.raises((capi.PcrError, ValueError)):
        pcr.KDTree(pts)


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under the package may reference it."""
    pkg = os.path.join(REPO, "point_cloud_registration_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", text, re.M), f
                code = re.sub(r"//.*|#.*", "", text)                  # comments may cite the oracle
                assert "libpcr_oracle" not in code and not re.search(r"\borc_\w+\s*\(", code), f
