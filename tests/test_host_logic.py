"""Host-side logic of the drop-in classes (no GPU): the Gauss-Newton driver's control flow with
the per-iteration sums supplied by the oracle, helpers, API surface."""

import inspect

import numpy as np
import pytest

import point_cloud_registration_amd as pcr
from point_cloud_registration_amd import _capi, math_tools as mt
from point_cloud_registration_amd.distributed import shard_bounds
from oracle import oracle as orc

KIND = {"icp": orc.ICP, "plane": orc.PLANE, "vplane": orc.VPLANE, "ndt": orc.NDT}


def pack29(H, g, e2, cnt):
    return np.concatenate([H[np.triu_indices(6)], g, [e2, cnt]])


class _OracleBacked:
    """Mixin: replaces the two GPU touch points of Registration with the CPU oracle so the HOST
    logic (align loop, quirks Q3/Q4/Q7, solve, plus) can be exercised without a GPU."""

    def __init__(self, *a, **kw):
        kw.setdefault("native_loop", False)       # the Python loop: the device-resident one needs a GPU
        super().__init__(*a, **kw)

    def _scan_for(self, source, fresh=False):
        return np.ascontiguousarray(source, dtype=np.float32)

    def _linearize(self, cur_T, scan):
        H, g, e2, cnt = orc.calc_H_g_e2(self.ORC_KIND, self._otarget, cur_T, scan, self._max_dist(),
                                        self._flags, with_count=True)
        out = pack29(H, g, e2, cnt)
        if self._comm is not None:
            out = self._comm.allreduce(out)
        H, g, e2, cnt = _capi.unpack29(out)
        self.last_correspondences = cnt
        return H, g, e2


def make(name, g, **kw):
    base = {"icp": pcr.ICP, "plane": pcr.PlaneICP, "vplane": pcr.VPlaneICP, "ndt": pcr.NDT}[name]
    cls = type("Oracle" + base.__name__, (_OracleBacked, base), {"ORC_KIND": KIND[name]})
    args = {"max_dist": float(g["max_dist"])}
    if name in ("vplane", "ndt"):
        args["voxel_size"] = float(g["voxel_size"])
    obj = cls(**args, **kw)
    if name in ("icp", "plane"):
        obj._otarget = orc.TargetPoints(g["target"], normals=g["plane_normals"])
    else:
        obj._otarget = orc.TargetVoxels(g["target"], float(g["voxel_size"]))
    return obj


@pytest.mark.parametrize("name", list(KIND))
def test_align_loop_matches_reference_trajectory(g2, name):
    obj = make(name, g2)
    with pytest.raises(ValueError, match="Target is not set"):
        obj.align(g2["source"])
    obj._is_target_set = True
    T = obj.align(g2["source"], np.eye(4))
    assert obj.last_iterations == g2[f"align_{name}_T"].shape[0]
    final = g2[f"align_{name}_final"]
    assert np.max(np.abs(T[:3, 3] - final[:3, 3])) < 1e-4
    ang = np.arccos(np.clip((np.trace(T[:3, :3] @ final[:3, :3].T) - 1) / 2, -1, 1))
    assert ang < 1e-4


def test_convergence_test_precedes_update():
    """Quirk Q4: when |dx| < tol the step is discarded (registration.py:106-111)."""
    class Fake(_OracleBacked, pcr.ICP):
        def _linearize(self, cur_T, scan):
            self.calls += 1
            return np.eye(6), -np.full(6, 1e-5), 0.0     # dx = 1e-5 * ones, |dx| < 1e-3
    f = Fake(); f.calls = 0; f._is_target_set = True
    T0 = np.eye(4); T0[0, 3] = 7.0
    T = f.align(np.zeros((4, 3), np.float32), T0)
    assert f.calls == 1 and np.array_equal(T, T0)


def test_max_iter_and_singular():
    class Fake(_OracleBacked, pcr.ICP):
        def _linearize(self, cur_T, scan):
            self.calls += 1
            return np.eye(6), -np.array([0.1, 0, 0, 0, 0, 0.0]), 1.0
    f = Fake(max_iter=4); f.calls = 0; f._is_target_set = True
    T = f.align(np.zeros((4, 3), np.float32))
    assert f.calls == 4 and abs(T[0, 3] - 0.4) < 1e-12 and f.last_iterations == 4

    class Zero(_OracleBacked, pcr.ICP):
        def _linearize(self, cur_T, scan):
            return np.zeros((6, 6)), np.zeros(6), 0.0
    z = Zero(); z._is_target_set = True
    with pytest.raises(np.linalg.LinAlgError):          # quirk Q7
        z.align(np.zeros((4, 3), np.float32))


def test_api_surface_matches_reference_names():
    """Same constructor keywords and defaults as the reference classes."""
    sig = inspect.signature
    assert list(sig(pcr.ICP.__init__).parameters)[:4] == ["self", "max_iter", "max_dist", "tol"]
    assert sig(pcr.ICP.__init__).parameters["max_iter"].default == 30
    assert sig(pcr.ICP.__init__).parameters["max_dist"].default == 2
    assert sig(pcr.ICP.__init__).parameters["tol"].default == 1e-3
    assert sig(pcr.PlaneICP.__init__).parameters["k"].default == 15
    assert list(sig(pcr.PlaneICP.set_target).parameters) == ["self", "target", "kdree", "norm"]
    assert list(sig(pcr.VPlaneICP.__init__).parameters)[:5] == ["self", "voxel_size", "max_iter", "max_dist", "tol"]
    assert sig(pcr.NDT.__init__).parameters["voxel_size"].default == 1.0
    assert sig(pcr.VoxelGrid.__init__).parameters["min_points"].default == 10
    assert list(sig(pcr.Registration.align).parameters) == ["self", "source", "init_T", "verbose"]
    for name in ["ICP", "PlaneICP", "VPlaneICP", "NDT", "KDTree", "VoxelGrid", "voxel_filter", "color_by_voxel",
                 "estimate_normals", "get_norm_lines", "estimate_norm_with_tree", "makeRt", "expSO3", "makeT",
                 "skews", "huber_weight", "plus", "transform_points", "skew_time_vector", "Registration"]:
        assert hasattr(pcr, name), name
    for cls in (pcr.ICP, pcr.PlaneICP, pcr.VPlaneICP, pcr.NDT):
        assert not cls().is_target_set()
        with pytest.raises(NotImplementedError):
            cls().update_target(None)


def test_math_helpers():
    rng = np.random.default_rng(1)
    v, w = rng.normal(size=(50, 3)), rng.normal(size=(50, 3))
    assert np.allclose(mt.skew_time_vector(v, w), np.cross(v, w))
    S = mt.skews(v)
    assert np.allclose(np.einsum("nij,nj->ni", S, w), np.cross(v, w))
    assert np.allclose(mt.skew(v[0]) @ w[0], np.cross(v[0], w[0]))
    assert np.allclose(mt.skew2(v), np.einsum("nji,njk->ik", S, S))
    T = mt.makeT(mt.expSO3(np.array([0.3, -0.2, 0.1])), np.array([1.0, 2.0, 3.0]))
    R, t = mt.makeRt(T)
    assert np.allclose(R @ R.T, np.eye(3)) and np.allclose(t, [1, 2, 3])
    p = rng.normal(size=(10, 3))
    assert np.allclose(mt.transform_points(T, p), p @ R.T + t)
    assert np.allclose(mt.huber_weight(np.array([0.5, 2.0, 4.0]), 1.0), [1.0, 0.5, 0.25])
    J = mt.numerical_derivative(lambda a: a ** 2, [np.array([1.0, 2.0])], 0)
    assert np.allclose(J, np.diag([2.0, 4.0]), atol=1e-4)


def test_voxel_host_utilities(g3):
    pts = g3["points_f32"]
    for vs in (0.5, 1.0):
        assert np.array_equal(pcr.get_keys(pts, vs), g3[f"f32_vs{vs}_keys"])
    c = pcr.color_by_voxel(pts, 0.5)
    assert c["xyz"].shape == pts.shape and c["irgb"].dtype == np.uint32
    lines = pcr.get_norm_lines(pts[:5], np.tile([0, 0, 1.0], (5, 1)).astype(np.float32), 0.1)
    assert lines.shape == (10, 3) and np.allclose(lines[1::2] - lines[0::2], [0, 0, 0.1])


def test_unpack29_and_sharding():
    rng = np.random.default_rng(0)
    A = rng.normal(size=(6, 6)); H = A + A.T
    out = pack29(H, np.arange(6.0), 3.5, 17)
    H2, g2, e2, cnt = _capi.unpack29(out)
    assert np.allclose(H2, H) and np.allclose(g2, np.arange(6.0)) and e2 == 3.5 and cnt == 17
    for n, w in ((10, 3), (7, 8), (1_060_000, 8), (0, 2)):
        spans = [shard_bounds(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1


# ----------------------------------------------------------------------------- round 3: seam hash, reuse policy
def test_hash64_content_hash():
    """pcr_hash64 (no GPU needed): deterministic, sensitive to one bit anywhere, independent of the number of
    pool threads (fixed 256 KiB chunks folded in order), and the class seam's digest separates C / F layouts."""
    import os
    import subprocess
    import sys
    rng = np.random.default_rng(0)
    a = rng.random((300_000, 3)).astype(np.float32)                 # 3.6 MB: 14 chunks, the parallel path
    h = _capi.hash64(a)
    assert h == _capi.hash64(a.copy()) and h == _capi.hash64(a)
    for idx in ((0, 0), (123_456, 1), (299_999, 2)):
        b = a.copy()
        b[idx] = np.nextafter(b[idx], np.float32(2.0))
        assert _capi.hash64(b) != h
    assert _capi.hash64(a[:1000].copy()) != _capi.hash64(a[:1001].copy())
    assert _capi.hash64(np.zeros((0, 3), np.float32)) == _capi.hash64(np.zeros((0, 3), np.float32))
    code = ("import sys, numpy as np; sys.path.insert(0, %r)\n"
            "from point_cloud_registration_amd import _capi\n"
            "a = np.random.default_rng(0).random((300000, 3)).astype(np.float32); print(_capi.hash64(a))\n"
            % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    for threads in ("0", "3"):
        out = subprocess.run([sys.executable, "-c", code], env={**os.environ, "PCR_HASH_THREADS": threads},
                             capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stderr[-500:]
        assert int(out.stdout.strip().splitlines()[-1]) == h, threads
    # the class seam: an F-ordered view (what (R @ P.T).T yields) is hashed in place, and differs from the C copy's key
    f = np.asfortranarray(a)
    dig = pcr.ICP._digest
    assert dig(f) == dig(np.asfortranarray(a.copy())) and dig(f) != dig(a)
    g = f.copy(order="F"); g[7, 1] += 1.0
    assert dig(g) != dig(f)


def test_hash64_survives_fork():
    """after fork() the child inherits the hash pool without its threads; the next >= 1 MiB hash must not
    wait for workers that do not exist.  (multiprocessing's default start method on Linux, DataLoader workers.)"""
    import os
    rng = np.random.default_rng(1)
    a = rng.random((300_000, 3)).astype(np.float32)
    h = _capi.hash64(a)                                            # the parent's pool exists now
    r, w = os.pipe()
    pid = os.fork()
    if pid == 0:                                                    # child
        try:
            os.close(r)
            import signal
            signal.alarm(30)                                        # a hang kills the child instead of the suite
            v = _capi.hash64(a)
            os.write(w, (b"ok" if v == h else b"bad"))
        finally:
            os._exit(0)
    os.close(w)
    got = os.read(r, 16)
    os.close(r)
    _, status = os.waitpid(pid, 0)
    assert got == b"ok", (got, status)
    assert _capi.hash64(a) == h                                     # and the parent's pool still works


def test_usable_cpus_is_the_quota():
    n = _capi.lib().pcr_usable_cpus()
    assert 1 <= n <= (os.cpu_count() or 1) if (os := __import__("os")) else True
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            assert n <= max(int(float(q) / float(per)), 1)
    except OSError:
        pass


def test_reuse_policy_restated():
    """gn_choose_nn_mode / gn_typical_motion (csrc/gn_math.h) restated: the automatic policy never starts tracking on a
    quadratically converging run, keeps going once tracking, and the typical displacement is what it says."""
    def typical_motion(Ta, Tb, c, e):
        pts = [np.array(c, float)]
        for ax in range(3):
            for sgn in (-1, 1):
                p = np.array(c, float); p[ax] += sgn * e[ax]; pts.append(p)
        return float(np.mean([np.linalg.norm((Tb[:3, :3] @ p + Tb[:3, 3]) - (Ta[:3, :3] @ p + Ta[:3, 3])) for p in pts]))

    def choose(reuse, have_prev, track_valid, motion, prev_motion, tau_len):
        if reuse == 0 or not have_prev:
            return 0
        if reuse == 2:
            return 2 if track_valid else 1
        if not (motion < tau_len):
            return 0
        if track_valid:
            return 2
        if motion == 0.0:
            return 1
        if 0.0 <= prev_motion < 8.0 * tau_len and motion > 0.3 * prev_motion:
            return 1
        return 0

    T1 = np.eye(4); T2 = np.eye(4); T2[:3, 3] = [0.003, 0.0, 0.004]
    assert abs(typical_motion(T1, T2, (0, 0, 0), (60, 30, 10)) - 0.005) < 1e-12       # a pure translation moves every probe alike
    R = mt.expSO3(np.array([0.0, 0.0, 1e-3])); T3 = np.eye(4); T3[:3, :3] = R
    m = typical_motion(T1, T3, (0, 0, 0), (60, 30, 10))
    assert 0.02 < m < 0.03                                                            # 1 mrad about z: 0 / 60 mm / 30 mm / 0 at the probes
    tau = 0.0125 * 0.405
    # plane_b01's measured steps (mm): 428, 236, 117, 18 -> never below tau, never tracking
    prev = -1.0
    for step in (0.428, 0.236, 0.117, 0.018):
        assert choose(1, True, False, step, prev, tau) == 0
        prev = step
    # a slow sub-millimetre tail starts tracking, then lists; a jump drops back to the plain search
    assert choose(1, True, False, 0.004, 0.006, tau) == 1 and choose(1, True, True, 0.003, 0.004, tau) == 2
    assert choose(1, True, True, 0.2, 0.003, tau) == 0 and choose(1, True, False, 0.0, 0.2, tau) == 1
    assert choose(1, True, False, 0.001, 0.02, tau) == 0                              # quadratic convergence: the loop ends first
    assert choose(0, True, True, 0.0, 0.0, tau) == 0 and choose(2, True, False, 1.0, 1.0, tau) == 1
    # ... and the C functions themselves (csrc/gn_math.h compiles as plain host C++) agree with the restatement
    import os
    import subprocess
    import tempfile
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = r"""
#include <stdio.h>
#include "gn_math.h"
int main() {
    double Ta[16] = {1,0,0,0, 0,1,0,0, 0,0,1,0, 0,0,0,1}, Tb[16];
    for (int i = 0; i < 16; ++i) Tb[i] = Ta[i];
    Tb[3] = 0.003; Tb[11] = 0.004;
    const float c[3] = {0, 0, 0}, e[3] = {60, 30, 10};
    printf("%.17g\n", gn_typical_motion(Ta, Tb, c, e));
    const double tau = 0.0125 * 0.405;
    const double cases[][5] = {{1,1,0,0.428,-1}, {1,1,0,0.018,0.117}, {1,1,0,0.004,0.006}, {1,1,1,0.003,0.004}, {1,1,1,0.2,0.003},
                               {1,1,0,0.0,0.2}, {1,1,0,0.001,0.02}, {0,1,1,0,0}, {2,1,0,1,1}, {2,1,1,1,1}, {1,0,1,0,0}};
    for (auto &k : cases) printf("%d\n", gn_choose_nn_mode((int)k[0], (int)k[1], (int)k[2], k[3], k[4], tau));
    return 0;
}
"""
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.cpp"), "w").write(src)
        subprocess.run(["g++", "-O1", "-I", os.path.join(repo, "point_cloud_registration_amd", "csrc"), os.path.join(d, "t.cpp"),
                        "-o", os.path.join(d, "t")], check=True, capture_output=True)
        out = subprocess.run([os.path.join(d, "t")], capture_output=True, text=True, check=True).stdout.split()
    assert abs(float(out[0]) - 0.005) < 1e-12
    want = [choose(1, True, False, 0.428, -1, tau), choose(1, True, False, 0.018, 0.117, tau), choose(1, True, False, 0.004, 0.006, tau),
            choose(1, True, True, 0.003, 0.004, tau), choose(1, True, True, 0.2, 0.003, tau), choose(1, True, False, 0.0, 0.2, tau),
            choose(1, True, False, 0.001, 0.02, tau), choose(0, True, True, 0, 0, tau), choose(2, True, False, 1, 1, tau),
            choose(2, True, True, 1, 1, tau), choose(1, False, True, 0, 0, tau)]
    assert [int(v) for v in out[1:]] == want


def test_gn_math_se3_against_the_reference_golden(g5):
    """csrc/gn_math.h compiled as plain host C++ (what api.hip's host loop and k_gn_update's device code share): expSO3 / plus
    against the REFERENCE's outputs of fixture g5 (math_tools.py:80-108, either side of the first-order branch, quirk Q3), and
    gn_sincos -- the one sin / cos both loops use since round 4 -- against libm over six decades of angles."""
    import os
    import subprocess
    import tempfile
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n = g5["omegas"].shape[0]
    rows = ",".join("{" + ",".join(repr(float(v)) for v in w) + "}" for w in g5["omegas"])
    dxs = ",".join("{" + ",".join(repr(float(v)) for v in w) + "}" for w in g5["dxs"])
    t0 = ",".join(repr(float(v)) for v in g5["T0"].reshape(16))
    src = r"""
#include <stdio.h>
#include <math.h>
#include "gn_math.h"
int main() {
    const double om[][3] = {%s};
    const double dx[][6] = {%s};
    const double T0[16] = {%s};
    for (int i = 0; i < %d; ++i) {
        double R[9]; gn_exp_so3(om[i], R);
        for (int k = 0; k < 9; ++k) printf("%%.17g ", R[k]);
        double T[16]; for (int k = 0; k < 16; ++k) T[k] = T0[k];
        gn_se3_plus(T, dx[i]);
        for (int k = 0; k < 16; ++k) printf("%%.17g ", T[k]);
        printf("\n");
    }
    double ws = 0, wc = 0;
    for (int i = 0; i < 400000; ++i) {
        const double x = 1e-3 * pow(1.00004, i);            /* 1e-3 .. 9e3 rad */
        double s, c; gn_sincos(x, &s, &c);
        ws = fmax(ws, fabs(s - sin(x))); wc = fmax(wc, fabs(c - cos(x)));
    }
    printf("%%.3e %%.3e\n", ws, wc);
    return 0;
}
""" % (rows, dxs, t0, n)
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.cpp"), "w").write(src)
        subprocess.run(["g++", "-O2", "-ffp-contract=off", "-I", os.path.join(repo, "point_cloud_registration_amd", "csrc"),
                        os.path.join(d, "t.cpp"), "-o", os.path.join(d, "t")], check=True, capture_output=True)
        out = subprocess.run([os.path.join(d, "t")], capture_output=True, text=True, check=True).stdout.strip().splitlines()
    for i in range(n):
        v = np.array([float(x) for x in out[i].split()])
        assert np.max(np.abs(v[:9].reshape(3, 3) - g5["Rs"][i])) < 1e-14, i
        assert np.max(np.abs(v[9:].reshape(4, 4) - g5["Ts"][i])) < 1e-13, i
    ws, wc = (float(x) for x in out[n].split())
    assert ws < 1e-12 and wc < 1e-12, (ws, wc)           # (~1 ulp below a few rad; the reduction loses digits at 1e3 rad)


def _filter_consts(maxabs, bound):
    """gn_filter_band / gn_filter_bounds (csrc/gn_math.h) through g++: the numbers the library computes."""
    import os
    import subprocess
    import tempfile
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = r"""
#include <stdio.h>
#include <stdlib.h>
#include "gn_math.h"
int main(int argc, char **argv) {
    const double band = gn_filter_band(atof(argv[1]));
    const float band_f = (float)(band * 1.000001);
    float b2, mu;
    gn_filter_bounds((double)band_f, atof(argv[2]), &b2, &mu);
    printf("%.9g %.9g %.9g\n", (double)band_f, (double)b2, (double)mu);
    return 0;
}
"""
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.cpp"), "w").write(src)
        subprocess.run(["g++", "-O1", "-I", os.path.join(repo, "point_cloud_registration_amd", "csrc"), os.path.join(d, "t.cpp"),
                        "-o", os.path.join(d, "t")], check=True, capture_output=True)
        out = subprocess.run([os.path.join(d, "t"), repr(float(maxabs)), repr(float(bound))], capture_output=True, text=True, check=True)
    return [np.float32(v) for v in out.stdout.split()]


@pytest.mark.parametrize("maxabs", [300.0, 3.0e4])
def test_centroid_filter_certificate(maxabs):
    """The inequality behind k_nn_filter (csrc/pass_device.h: nn_point_filter), restated on the CPU with the library's own
    constants: a float32 search over the ROUNDED centroids nominates a winner and reports a lower bound on the float32-space
    distance of every other rounded centroid; whenever `(lbq * 0.99999 - band)^2 > d64 * 1.000001` holds, the float64 search
    (distance, then smaller index) must return that same winner -- for ordinary clouds, near-ties from 1e-7 to 1e-3 m,
    exact duplicates, coordinates at the edge of the box.  The restatement grants the device the LARGEST bound it could
    report (the true second-smallest distance, capped by the tracking margin), i.e. the most permissive certificate."""
    rng = np.random.default_rng(int(maxabs))
    band, bound2_ff, mu = _filter_consts(maxabs, 2.0 * (1 + 1e-6))
    assert band <= 0.01 * 1.0 or maxabs > 1e4            # (1 m cells: the filter is built at 300 m; 3e4 m is a 0.5 % case)
    n_c, n_q = 4000, 3000
    base = rng.uniform(-1, 1, 3) * (maxabs - 40.0)
    cent = base + rng.uniform(-20, 20, (n_c, 3))
    cent[:, 2] = base[2] + rng.uniform(-2, 2, n_c)
    q = (base + rng.uniform(-20, 20, (n_q, 3)) * [1, 1, 0.1]).astype(np.float32)
    # adversarial pairs: for the first 600 queries, two centroids at distances r and r + eps in random directions
    k = 600
    for i in range(k):
        r = rng.uniform(0.01, 1.5); eps = 10.0 ** rng.uniform(-7, -3)
        u, v = rng.normal(size=3), rng.normal(size=3)
        cent[2 * i] = q[i].astype(np.float64) + u / np.linalg.norm(u) * r
        cent[2 * i + 1] = q[i].astype(np.float64) + v / np.linalg.norm(v) * (r + eps)
    cent[3000:3050] = cent[2000:2050]                  # exact duplicates
    c32 = cent.astype(np.float32)
    assert np.max(np.linalg.norm(c32.astype(np.float64) - cent, axis=1)) <= float(band)
    n_cert = n_pend = 0
    for i in range(n_q):
        d32 = ((q[i] - c32) ** 2).sum(1, dtype=np.float32)                     # float32 arithmetic, like dist2_f32 up to an ulp
        order = np.lexsort((np.arange(n_c), d32))
        w, best, second = order[0], d32[order[0]], d32[order[1]]
        if not best < bound2_ff:
            continue                                                            # nothing within the bound: certified "none"
        lb2q = min(np.float32(second), (np.sqrt(best) + mu) ** 2)               # min(second, pmin)
        lbq = np.float32(np.sqrt(np.float32(lb2q))) * np.float32(0.99999) - band
        dd = q[i].astype(np.float64) - cent
        d64 = (dd[:, 0] * dd[:, 0] + dd[:, 1] * dd[:, 1]) + dd[:, 2] * dd[:, 2]
        cert = lbq > 0 and float(lbq) * float(lbq) > d64[w] * 1.000001
        if cert:
            n_cert += 1
            exact = np.lexsort((np.arange(n_c), d64))[0]
            assert exact == w, (i, w, exact, d64[w], d64[exact], float(lbq))
            assert np.sort(d64)[1] > d64[w]                                      # strictly closer than everything else
        else:
            n_pend += 1
    assert n_cert > 0.7 * n_q and n_pend >= 1                                   # the certificate is not vacuous, nor always true
