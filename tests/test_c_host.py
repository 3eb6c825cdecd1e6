"""The C ABI from plain C99 (examples/c_host.c): the header compiles as strict C, the library links
without Python, and a C host gets the same numbers as the ctypes binding."""

import os
import subprocess

import numpy as np
import pytest

from conftest import REPO

PKG = os.path.join(REPO, "point_cloud_registration_amd")
SRC = os.path.join(REPO, "examples", "c_host.c")


def _build(tmp_path):
    exe = str(tmp_path / "c_host")
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-O2", "-I", os.path.join(REPO, "include"),
           SRC, "-o", exe, "-L", PKG, "-lpcr_hip", f"-Wl,-rpath,{PKG}", "-Wl,-rpath-link,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def _env():
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    return env


def test_header_is_strict_c99_and_library_links(tmp_path):
    if not os.path.exists(os.path.join(PKG, "libpcr_hip.so")):
        pytest.skip("libpcr_hip.so not built")
    exe = _build(tmp_path)
    r = subprocess.run([exe, "version"], capture_output=True, text=True, env=_env())
    assert r.returncode == 0, r.stderr
    assert r.stdout.startswith("pcr-hip") and "GPU(s) visible" in r.stdout
    r = subprocess.run([exe], capture_output=True, text=True, env=_env())
    assert r.returncode == 2 and "usage" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("kind", [0, 1, 2, 3])
def test_c_host_matches_ctypes_binding(tmp_path, kind):
    from point_cloud_registration_amd import _capi
    from point_cloud_registration_amd.synthetic import perturbed_scan, street
    target = street(400_000, seed=5)
    scan, _ = perturbed_scan(target, 50_000, seed=6)
    target.astype(np.float32).tofile(tmp_path / "target.f32")
    scan.astype(np.float32).tofile(tmp_path / "scan.f32")
    exe = _build(tmp_path)
    r = subprocess.run([exe, "run", str(kind), str(tmp_path / "target.f32"), str(tmp_path / "scan.f32"), "1.0"],
                       capture_output=True, text=True, env=_env(), timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = {l.split()[0]: l.split()[1:] for l in r.stdout.splitlines() if l.startswith(("linearize", "align"))}
    lin_c = np.array(lines["linearize"], dtype=np.float64)
    iters_c, T_c = int(lines["align"][0]), np.array(lines["align"][1:], dtype=np.float64).reshape(4, 4)

    ctx = _capi.get_context(0)
    if kind in (_capi.ICP, _capi.PLANE):
        tgt = _capi.Target.points(ctx, target)
        if kind == _capi.PLANE:
            tgt.estimate_normals(15, compat=True, want=False)
    else:
        tgt = _capi.Target.voxels(ctx, target, 1.0, 10)
    sc = _capi.Scan(ctx, scan)
    lin_py = _capi.linearize(tgt, sc, kind, np.eye(4), 2.0)
    T_py, iters_py = _capi.align(tgt, sc, kind, np.eye(4), 30, 1e-3, 2.0)
    assert lin_c.shape == (29,) and lin_c[28] > 10_000      # most of the scan finds a correspondence
    assert np.array_equal(lin_c, lin_py)              # same library, same order of operations: bit-identical
    assert iters_c == iters_py and np.array_equal(T_c, T_py), (iters_c, iters_py, np.abs(T_c - T_py).max())
