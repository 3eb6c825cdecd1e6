"""SE(3)/SO(3) host helpers with the reference's names and semantics.

These are O(1) (or utility) host-side functions: the Gauss-Newton update, the boxplus and
the tolerance test stay on the host exactly as in the reference
(``point_cloud_registration/math_tools.py:22-113``); the O(N) work they sit next to runs
in the HIP kernels.  Quirks kept on purpose:

* Q3 -- ``expSO3`` returns the first-order ``I + skew(w)`` when ``w.w <= 1e-5``
  (reference ``math_tools.py:12,88-91``).
* Q2 -- ``plus`` is right-multiplicative: ``T @ [exp(w), v; 0, 1]`` (``math_tools.py:101-108``).
"""

import numpy as np

epsilon = 1e-5


def huber_weight(r, d=1.0):
    """Huber weights ``min(1, d / r)`` (reference ``math_tools.py:15-19``; unused by align)."""
    r = np.asarray(r)
    w = np.ones_like(r)
    big = r > d
    w[big] = d / r[big]
    return w


def skew(vector):
    """3x3 cross-product matrix of one vector (reference ``math_tools.py:61-64``)."""
    x, y, z = vector[0], vector[1], vector[2]
    return np.array([[0, -z, y],
                     [z, 0, -x],
                     [-y, x, 0]])


def skews(vectors):
    """Batch of cross-product matrices, shape (N, 3, 3) float64 (``math_tools.py:34-41``)."""
    v = np.asarray(vectors)
    out = np.zeros((v.shape[0], 3, 3))
    out[:, 0, 1] = -v[:, 2]
    out[:, 0, 2] = v[:, 1]
    out[:, 1, 0] = v[:, 2]
    out[:, 1, 2] = -v[:, 0]
    out[:, 2, 0] = -v[:, 1]
    out[:, 2, 1] = v[:, 0]
    return out


def skew2(v):
    """sum_i skew(v_i)^T skew(v_i) from the six second moments (``math_tools.py:44-58``)."""
    v = np.asarray(v)
    m = v.T @ v
    tr = np.trace(m)
    return tr * np.eye(3) - m


def skew_time_vector(v1, v2):
    """Row-wise ``skew(v1_i) @ v2_i`` = ``v1_i x v2_i`` (``math_tools.py:22-31``)."""
    return np.cross(np.asarray(v1, dtype=np.float64), np.asarray(v2, dtype=np.float64))


def makeT(R, t):
    n = t.shape[0]
    T = np.eye(n + 1)
    T[:n, :n] = R
    T[:n, n] = t
    return T


def makeRt(T):
    n = T.shape[0] - 1
    return T[:n, :n], T[:n, n]


def expSO3(omega):
    """Rodrigues' formula with the reference's first-order branch (quirk Q3)."""
    omega = np.asarray(omega, dtype=np.float64)
    theta2 = float(omega @ omega)
    W = skew(omega)
    if theta2 <= epsilon:
        return np.eye(3) + W
    theta = np.sqrt(theta2)
    K = W / theta
    return np.eye(3) + np.sin(theta) * K + (1.0 - np.cos(theta)) * (K @ K)


def plus(T, dx):
    """Boxplus on SE(3): ``T @ makeT(expSO3(dx[3:]), dx[:3])`` (quirk Q2)."""
    dx = np.asarray(dx, dtype=np.float64)
    return T @ makeT(expSO3(dx[3:]), dx[:3])


def transform_points(T, points):
    """``(R @ P.T).T + t`` in the dtype NumPy promotes to (``math_tools.py:111-113``)."""
    R, t = makeRt(T)
    return (R @ points.T).T + t


def numerical_derivative(func, param, idx, plus=lambda a, b: a + b,
                         minus=lambda a, b: a - b, delta=1e-5):
    """Forward-difference Jacobian helper (``math_tools.py:116-129``; test utility)."""
    r0 = func(*param)
    n = param[idx].shape[0]
    J = np.zeros((r0.shape[0], n))
    for j in range(n):
        step = np.zeros(n)
        step[j] = delta
        shifted = list(param)
        shifted[idx] = plus(param[idx], step)
        J[:, j] = minus(func(*shifted), r0) / delta
    return J
