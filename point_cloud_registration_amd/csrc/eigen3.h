// Symmetric 3x3 eigen-solver and closed-form inverse used at set_target time (device).
#pragma once

#include <hip/hip_runtime.h>

// Cyclic Jacobi in float64.  Returns in n[3] the unit eigenvector of the SMALLEST eigenvalue
// (numpy.linalg.eigh's eigenvectors[:, 0], reference voxel.py:157-158 and
// estimate_normals.py:75-76; the sign of an eigenvector is arbitrary in both).
__device__ __forceinline__ void smallest_eigvec3(const double c[6] /* xx xy xz yy yz zz */, double n[3]) {
    double a[3][3] = {{c[0], c[1], c[2]}, {c[1], c[3], c[4]}, {c[2], c[4], c[5]}};
    double v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 60; ++sweep) {
        const double off = fabs(a[0][1]) + fabs(a[0][2]) + fabs(a[1][2]);
        const double dia = fabs(a[0][0]) + fabs(a[1][1]) + fabs(a[2][2]);
        if (off <= 1e-300 || off <= 1e-18 * dia) break;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
#pragma unroll
            for (int q = p + 1; q < 3; ++q) {
                if (a[p][q] == 0.0) continue;
                const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const double akp = a[k][p], akq = a[k][q];
                    a[k][p] = cs * akp - sn * akq; a[k][q] = sn * akp + cs * akq;
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const double apk = a[p][k], aqk = a[q][k];
                    a[p][k] = cs * apk - sn * aqk; a[q][k] = sn * apk + cs * aqk;
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const double vkp = v[k][p], vkq = v[k][q];
                    v[k][p] = cs * vkp - sn * vkq; v[k][q] = sn * vkp + cs * vkq;
                }
            }
        }
    }
    // column of the smallest diagonal entry, selected without dynamic register indexing
    const double e0 = a[0][0], e1 = a[1][1], e2 = a[2][2];
    const bool s1 = e1 < e0, s2 = e2 < (s1 ? e1 : e0);
    double x = s2 ? v[0][2] : (s1 ? v[0][1] : v[0][0]);
    double y = s2 ? v[1][2] : (s1 ? v[1][1] : v[1][0]);
    double z = s2 ? v[2][2] : (s1 ? v[2][1] : v[2][0]);
    const double nrm = sqrt(x * x + y * y + z * z);
    n[0] = x / nrm; n[1] = y / nrm; n[2] = z / nrm;
}

// voxel.py:69-102 calc_icov: adjugate / determinant, the same operation order; det == 0 -> 1e6.
__device__ __forceinline__ void icov_closed_form(const double m[9], double o[9]) {
    const double a = m[0], b = m[4], c = m[8], d = m[1], e = m[2], f = m[5];
    const double f2 = f * f, d2 = d * d, e2 = e * e;
    const double bc = b * c, ac = a * c, ab = a * b;
    const double dc = d * c, de = d * e, ef = e * f;
    const double af = a * f, df = d * f, eb = e * b;
    double det = a * bc + 2 * de * f - a * f2 - b * e2 - c * d2;
    if (det == 0) det = 1000000;
    const double c0 = (bc - f2) / det, c1 = -(dc - ef) / det, c2 = (df - eb) / det;
    const double c3 = (ac - e2) / det, c4 = -(af - de) / det, c5 = (ab - d2) / det;
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c1; o[4] = c3; o[5] = c4; o[6] = c2; o[7] = c4; o[8] = c5;
}
