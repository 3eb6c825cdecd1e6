"""Synthetic clouds used wherever the reference uses ``data/B-01.pcd``.

``B-01.pcd`` is absent from the reference checkout (``.MISSING_LARGE_BLOBS``) and there
is no network, so every benchmark and large-scale test runs on the street-like stand-in
that SURVEY.md section 8(d) defines.  Scan generation mirrors the semantics of the
reference harness (``benchmark/test_data.py:21-44``: rigid transform of the map, random
subsample without replacement, N(0, 0.005) noise) without reusing its code.
"""

import numpy as np

from .math_tools import expSO3


def street(n, seed=0, center=(0.0, 0.0)):
    """Street-like cloud of ``n`` float32 points.

    50 % ground plane (z ~ N(0, 0.02)) over x in U(-60, 60), y in U(-30, 30);
    40 % four vertical walls (y = +-30, x = +-60, z in U(0, 20), wall-normal coordinate
    + N(0, 0.02)); 10 % clutter with x, y uniform and z in U(0, 5).
    """
    rng = np.random.default_rng(seed)
    n = int(n)
    n_ground = n // 2
    n_wall = (n * 4) // 10
    n_clutter = n - n_ground - n_wall
    pts = np.empty((n, 3), dtype=np.float64)

    g = pts[:n_ground]
    g[:, 0] = rng.uniform(-60.0, 60.0, n_ground)
    g[:, 1] = rng.uniform(-30.0, 30.0, n_ground)
    g[:, 2] = rng.normal(0.0, 0.02, n_ground)

    w = pts[n_ground:n_ground + n_wall]
    which = rng.integers(0, 4, n_wall)
    along_x = rng.uniform(-60.0, 60.0, n_wall)
    along_y = rng.uniform(-30.0, 30.0, n_wall)
    off = rng.normal(0.0, 0.02, n_wall)
    w[:, 2] = rng.uniform(0.0, 20.0, n_wall)
    ywall = which < 2
    w[ywall, 0] = along_x[ywall]
    w[ywall, 1] = np.where(which[ywall] == 0, -30.0, 30.0) + off[ywall]
    xwall = ~ywall
    w[xwall, 0] = np.where(which[xwall] == 2, -60.0, 60.0) + off[xwall]
    w[xwall, 1] = along_y[xwall]

    c = pts[n_ground + n_wall:]
    c[:, 0] = rng.uniform(-60.0, 60.0, n_clutter)
    c[:, 1] = rng.uniform(-30.0, 30.0, n_clutter)
    c[:, 2] = rng.uniform(0.0, 5.0, n_clutter)

    pts[:, 0] += center[0]
    pts[:, 1] += center[1]
    return pts.astype(np.float32)


def street_normals(pts, center=(0.0, 0.0)):
    """Analytic unit normals of a ``street`` cloud, from the coordinates alone: the first half of the points
    are the ground (0, 0, 1); of the next 40 % those within 0.2 m of y = +-30 belong to the y-walls (0, 1, 0),
    the others to the x-walls (1, 0, 0); the clutter gets (0, 0, 1).  A reproducible, geometrically meaningful
    input for ``PlaneICP.set_target(target, tree, norm)`` (plane_icp.py:25-27) that does not depend on any
    normal estimator."""
    n = pts.shape[0]
    n_ground = n // 2
    n_wall = (n * 4) // 10
    out = np.zeros((n, 3), dtype=np.float32)
    out[:, 2] = 1.0
    w = slice(n_ground, n_ground + n_wall)
    ywall = np.abs(np.abs(pts[w, 1] - np.float32(center[1])) - 30.0) < 0.2
    out[w, 2] = 0.0
    out[w, 1] = ywall
    out[w, 0] = ~ywall
    return out


def street_tiled(n_total, seed=0, per_tile=1_000_000):
    """Constant-density large cloud: tiles of the 120 x 60 m street on a near-square grid
    with per-tile seeds, mean-centred."""
    n_total = int(n_total)
    n_tiles = max(1, int(round(n_total / per_tile)))
    ny = max(1, int(np.floor(np.sqrt(n_tiles * 2.0))))   # tiles are 120 x 60: 2 rows per column width
    while n_tiles % ny:
        ny -= 1
    nx = n_tiles // ny
    base = n_total // n_tiles
    out = np.empty((n_total, 3), dtype=np.float32)
    pos = 0
    for t in range(n_tiles):
        cnt = base + (1 if t < n_total - base * n_tiles else 0)
        ix, iy = t % nx, t // nx
        out[pos:pos + cnt] = street(cnt, seed=seed * 100003 + t, center=(ix * 120.0, iy * 60.0))
        pos += cnt
    out -= out.mean(axis=0, dtype=np.float64).astype(np.float32)
    return out


def street_tiled_normals(pts, per_tile=1_000_000):
    """Analytic unit normals of a ``street_tiled`` cloud (float32, axis-aligned: exactly representable), from the point order
    and the coordinates alone -- tile by tile what ``street_normals`` does for one street, with each tile's centre recovered
    from the mid-range of its own ground points (``street_tiled`` subtracts the global mean afterwards).  The reproducible
    ``PlaneICP.set_target(target, tree, norm)`` input (plane_icp.py:25-27) of the 1e8-point reference fixture g13."""
    n_total = pts.shape[0]
    n_tiles = max(1, int(round(n_total / per_tile)))
    base = n_total // n_tiles
    out = np.zeros((n_total, 3), dtype=np.float32)
    pos = 0
    for t in range(n_tiles):
        cnt = base + (1 if t < n_total - base * n_tiles else 0)
        tile = pts[pos:pos + cnt]
        n_ground = cnt // 2
        n_wall = (cnt * 4) // 10
        g = tile[:n_ground]
        cy = 0.5 * (float(g[:, 1].min()) + float(g[:, 1].max()))
        o = out[pos:pos + cnt]
        o[:n_ground, 2] = 1.0
        w = tile[n_ground:n_ground + n_wall]
        ywall = np.abs(np.abs(w[:, 1].astype(np.float64) - cy) - 30.0) < 0.2
        o[n_ground:n_ground + n_wall, 1] = ywall
        o[n_ground:n_ground + n_wall, 0] = ~ywall
        o[n_ground + n_wall:, 2] = 1.0
        pos += cnt
    return out


def lidar_sweep(n, seed=0, sensor=(-20.0, 5.0, 1.8), rings=64, elev_deg=(-24.8, 15.0),
                range_limits=(0.5, 150.0), noise=0.02):
    """One revolution of a spinning ``rings``-beam LiDAR standing in the ``street`` geometry (ground z = 0 over
    x in [-60, 60], y in [-30, 30]; walls y = +-30 and x = +-60 up to z = 20): ``n`` float32 returns.

    Unlike ``street`` (constant density) this is what a real scan -- and the reference's absent B-01 street scan
    (``data/README.md:1-8``) -- looks like to a spatial index: point density falls like 1/r^2 with the range, the
    ground is a set of concentric ring lines (hundreds of returns per metre of arc at 4 m, a few at 60 m), and most
    of the space between the rings is empty.  Beams that leave the box through the open top give no return; azimuths
    are drawn until ``n`` returns exist.  Range noise N(0, ``noise``) along the beam.
    """
    rng = np.random.default_rng(seed)
    n = int(n)
    s = np.asarray(sensor, dtype=np.float64)
    elev = np.deg2rad(np.linspace(elev_deg[0], elev_deg[1], rings))
    out = np.empty((0, 3), dtype=np.float64)
    per_ring = max(16, int(np.ceil(n / rings * 1.6)))
    while out.shape[0] < n:
        az = (np.arange(per_ring) + rng.uniform(0.0, 1.0)) * (2.0 * np.pi / per_ring)
        a, e = np.meshgrid(az, elev, indexing="ij")             # azimuth-major: returns interleave the rings like a real sweep
        a = a.ravel() + rng.normal(0.0, 2.0e-4, a.size)
        e = e.ravel()
        d = np.stack([np.cos(e) * np.cos(a), np.cos(e) * np.sin(a), np.sin(e)], axis=1)
        with np.errstate(divide="ignore", invalid="ignore"):
            t_ground = np.where(d[:, 2] < 0, -s[2] / d[:, 2], np.inf)
            t_x = np.where(d[:, 0] > 0, (60.0 - s[0]) / d[:, 0], np.where(d[:, 0] < 0, (-60.0 - s[0]) / d[:, 0], np.inf))
            t_y = np.where(d[:, 1] > 0, (30.0 - s[1]) / d[:, 1], np.where(d[:, 1] < 0, (-30.0 - s[1]) / d[:, 1], np.inf))
        t = np.minimum(t_ground, np.minimum(t_x, t_y))
        hit = s + d * t[:, None]
        ok = np.isfinite(t) & (t >= range_limits[0]) & (t <= range_limits[1]) & (hit[:, 2] <= 20.0) & (hit[:, 2] >= -1e-9)
        t = t[ok] + rng.normal(0.0, noise, int(ok.sum()))
        out = np.concatenate([out, s + d[ok] * t[:, None]])
    return np.ascontiguousarray(out[:n], dtype=np.float32)


def lidar_normals(pts):
    """Analytic unit normals of a ``lidar_sweep`` cloud from the coordinates alone (the surfaces are the street's: ground z = 0,
    walls y = +-30 and x = +-60): a reproducible input for ``PlaneICP.set_target(target, tree, norm)`` (plane_icp.py:25-27)."""
    out = np.zeros((pts.shape[0], 3), dtype=np.float32)
    ywall = np.abs(np.abs(pts[:, 1]) - 30.0) < 0.5
    xwall = (np.abs(np.abs(pts[:, 0]) - 60.0) < 0.5) & ~ywall
    out[:, 2] = ~(ywall | xwall)
    out[:, 1] = ywall
    out[:, 0] = xwall
    return out


def make_T(so3, t):
    T = np.eye(4)
    T[:3, :3] = expSO3(np.asarray(so3, dtype=np.float64))
    T[:3, 3] = np.asarray(t, dtype=np.float64)
    return T


# perturbation used by BASELINE.json configs 2-5
T_TRUE_SO3 = (0.01, -0.02, 0.015)
T_TRUE_T = (0.05, 0.02, -0.1)


def harness_scan(target, num_points=100_000, so3=(0.0, 0.0, 0.0), t=(0.0, 0.0, 0.3),
                 noise=0.005, seed=1):
    """Scan in the reference harness' style: ``scan = R map + t``, random subsample without
    replacement, Gaussian noise.  ``align(scan, I)`` then recovers roughly the inverse."""
    rng = np.random.default_rng(seed)
    R = expSO3(np.asarray(so3, dtype=np.float64))
    scan = (R @ target.astype(np.float64).T).T + np.asarray(t, dtype=np.float64)
    num_points = min(int(num_points), scan.shape[0])
    idx = rng.choice(scan.shape[0], num_points, replace=False)
    scan = scan[idx]
    scan += rng.normal(0.0, noise, scan.shape)
    return np.ascontiguousarray(scan, dtype=np.float32)


def perturbed_scan(target, num_points=None, noise=0.005, seed=2,
                   so3=T_TRUE_SO3, t=T_TRUE_T):
    """Scan = T_true^-1 applied to the (sub-sampled) target + noise; ``align`` recovers T_true."""
    rng = np.random.default_rng(seed)
    T = make_T(so3, t)
    Rinv = T[:3, :3].T
    tinv = -Rinv @ T[:3, 3]
    pts = target
    if num_points is not None and num_points < target.shape[0]:
        idx = rng.choice(target.shape[0], int(num_points), replace=False)
        pts = target[idx]
    scan = (Rinv @ pts.astype(np.float64).T).T + tinv
    scan += rng.normal(0.0, noise, scan.shape)
    return np.ascontiguousarray(scan, dtype=np.float32), T      # (C order: (R @ P.T).T is a transposed view)
