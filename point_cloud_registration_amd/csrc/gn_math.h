// The O(1) tail of a Gauss-Newton iteration, shared by the host driver (api.hip) and the one-wave kernel
// k_gn_update (kernels.hip): dx = -solve(H, g), the |dx| < tol test, T <- plus(T, dx).
// Reference: registration.py:103-111 (loop body), math_tools.py:80-108 (expSO3, plus).
#pragma once

#include <math.h>

#if defined(__HIPCC__)
#define PCR_HD __host__ __device__
#else
#define PCR_HD
#endif

// numpy.linalg.solve: LU with partial pivoting, exact-zero pivot = singular (quirk Q7).
// A is caller-provided 6 x 7 working storage.  Every loop has compile-time bounds and the row exchange is
// written as a predicated swap against each candidate row, so that on the device the whole system stays in
// registers (a run-time row index would send it through LDS / scratch: ~10 us for this one thread instead
// of ~1); the arithmetic and its order are those of the textbook loop.
PCR_HD static inline int gn_solve6(double (*A)[7], const double H[36], const double g[6], double x[6]) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
#pragma unroll
        for (int j = 0; j < 6; ++j) A[i][j] = H[6 * i + j];
        A[i][6] = g[i];
    }
    int singular = 0;
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        int piv = c;
        double pmax = fabs(A[c][c]);
#pragma unroll
        for (int r = c + 1; r < 6; ++r) { const double v = fabs(A[r][c]); if (v > pmax) { pmax = v; piv = r; } }
        if (pmax == 0.0) singular = 1;
#pragma unroll
        for (int r = c + 1; r < 6; ++r) {
            const bool sw = piv == r;
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                const double a = A[c][j], b = A[r][j];
                A[c][j] = sw ? b : a; A[r][j] = sw ? a : b;
            }
        }
        if (!singular) {
#pragma unroll
            for (int r = c + 1; r < 6; ++r) {
                const double f = A[r][c] / A[c][c];
#pragma unroll
                for (int j = c; j < 7; ++j) A[r][j] -= f * A[c][j];
            }
        }
    }
    if (singular) return 1;
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        double v = A[i][6];
#pragma unroll
        for (int j = i + 1; j < 6; ++j) v -= A[i][j] * x[j];
        x[i] = v / A[i][i];
    }
    return 0;
}

// sin and cos of one angle from ONE piece of code for host and device (round 4).  libm's and the device library's sin / cos
// may differ in the last bit, and the full Rodrigues branch below then makes the host-driven and the device-resident loop
// part ways after a large step (found by tools/soak.py on a one-voxel target: 24 correspondences, an ill-conditioned H, a
// 0.3 rad step).  Cody-Waite reduction by pi/2 (three-part constant) + the fdlibm kernel polynomials, plain multiplies and
// adds (the TUs are compiled with -ffp-contract=off): within ~1 ulp of libm, identical wherever it is compiled.
PCR_HD static inline void gn_sincos(double x, double *sn, double *cs) {
    const double k = floor(x * 6.36619772367581382433e-01 + 0.5);            // x * 2/pi, nearest
    double r = x - k * 1.57079632673412561417e+00;                             // pi/2 in three parts (33 + 33 + 53 bits)
    r = r - k * 6.07710050630396597660e-11;
    r = r - k * 2.02226624871116645580e-21;
    const double z = r * r;
    const double ps = -1.66666666666666324348e-01 + z * (8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 +
                      z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10))));
    const double pc = 4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 +
                      z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11))));
    const double s0 = r + r * z * ps;
    const double c0 = (1.0 - 0.5 * z) + z * z * pc;
    const int q = (int)(k - 4.0 * floor(k * 0.25));                            // k mod 4 (k >= 0 here, fine for k < 0 too)
    *sn = q == 0 ? s0 : (q == 1 ? c0 : (q == 2 ? -s0 : -c0));
    *cs = q == 0 ? c0 : (q == 1 ? -s0 : (q == 2 ? -c0 : s0));
}

// math_tools.py:80-98: first-order I + skew(w) when w.w <= 1e-5 (quirk Q3), Rodrigues otherwise
PCR_HD static inline void gn_exp_so3(const double w[3], double R[9]) {
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    const double W[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
    if (th2 <= 1e-5) {
        for (int i = 0; i < 9; ++i) R[i] = W[i];
    } else {
        const double th = sqrt(th2);
        double sn, cth;
        gn_sincos(th, &sn, &cth);
        const double omc = 1.0 - cth;
        double K[9];
        for (int i = 0; i < 9; ++i) K[i] = W[i] / th;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                double kk = 0;
                for (int k = 0; k < 3; ++k) kk += K[3 * i + k] * K[3 * k + j];
                R[3 * i + j] = sn * K[3 * i + j] + omc * kk;
            }
    }
    R[0] += 1; R[4] += 1; R[8] += 1;
}

// math_tools.py:101-108: T <- T @ [exp(w), v; 0 1] (quirk Q2)
PCR_HD static inline void gn_se3_plus(double T[16], const double dx[6]) {
    double dR[9];
    gn_exp_so3(dx + 3, dR);
    const double D[16] = {dR[0], dR[1], dR[2], dx[0], dR[3], dR[4], dR[5], dx[1], dR[6], dR[7], dR[8], dx[2], 0, 0, 0, 1};
    double r[16];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double v = 0;
            for (int k = 0; k < 4; ++k) v += T[4 * i + k] * D[4 * k + j];
            r[4 * i + j] = v;
        }
    for (int i = 0; i < 16; ++i) T[i] = r[i];
}

// One Gauss-Newton step from the 29 sums (include/pcr.h layout).  Returns 0 = stepped (T updated),
// 1 = converged (|dx| < tol: the test precedes the update, quirk Q4; T unchanged), 2 = singular.
PCR_HD static inline int gn_step(double (*A)[7], const double o[29], double tol, double T[16]) {
    double H[36], g[6], dx[6];
    int p = 0;
    for (int i = 0; i < 6; ++i) for (int j = i; j < 6; ++j) { H[6 * i + j] = o[p]; H[6 * j + i] = o[p]; ++p; }
    for (int i = 0; i < 6; ++i) g[i] = o[21 + i];
    if (gn_solve6(A, H, g, dx)) return 2;
    double nrm = 0;
    for (int i = 0; i < 6; ++i) { dx[i] = -dx[i]; nrm += dx[i] * dx[i]; }
    if (sqrt(nrm) < tol) return 1;
    gn_se3_plus(T, dx);
    return 0;
}

// ---- certified reuse: when is it worth trying? -------------------------------------------------------
// Typical displacement of a scan between two poses: mean over the centre and the six face centres of its
// bounding box (centre c, half extents e) of |Tb p - Ta p|.  The same code decides on the host
// (pcr_linearize) and on the device (k_gn_update), so both loops take the same decisions -- which only
// affect speed: the certificate itself is exact, a pass returns the same bits whatever was decided.
PCR_HD static inline double gn_typical_motion(const double Ta[16], const double Tb[16], const float c[3], const float e[3]) {
    double sum = 0;
    for (int k = 0; k < 7; ++k) {
        double p[3] = {(double)c[0], (double)c[1], (double)c[2]};
        if (k > 0) { const int ax = (k - 1) >> 1; p[ax] += ((k - 1) & 1) ? (double)e[ax] : -(double)e[ax]; }
        double d2 = 0;
        for (int i = 0; i < 3; ++i) {
            const double a = Ta[4 * i] * p[0] + Ta[4 * i + 1] * p[1] + Ta[4 * i + 2] * p[2] + Ta[4 * i + 3];
            const double b = Tb[4 * i] * p[0] + Tb[4 * i + 1] * p[1] + Tb[4 * i + 2] * p[2] + Tb[4 * i + 3];
            d2 += (b - a) * (b - a);
        }
        sum += sqrt(d2);
    }
    return sum / 7.0;
}

// 0 = plain search, 1 = tracking search (leaves certifiable matches behind), 2 = certify the previous matches and
// search only the rest (PCR_NN_FULL / TRACK / LIST in pcr_internal.h).  reuse: 0 off, 1 automatic, 2 forced.
// Automatic: nothing while the scan still moves by more than tau_len per pass.  Below that, keep going once
// tracking; START tracking (a tracking pass costs 10-40 % more than a plain one and pays only if further passes
// follow) when the pose is being re-evaluated in place (motion 0) or converges slowly (this step more than 0.3 of
// the previous one: with the quadratic convergence of a well-conditioned Gauss-Newton run the loop ends first).
PCR_HD static inline int gn_choose_nn_mode(int reuse, int have_prev, int track_valid, double motion, double prev_motion,
                                           double tau_len) {
    if (reuse == 0 || !have_prev) return 0;
    if (reuse == 2) return track_valid ? 2 : 1;
    if (!(motion < tau_len)) return 0;
    if (track_valid) return 2;
    if (motion == 0.0) return 1;
    if (prev_motion >= 0.0 && prev_motion < 8.0 * tau_len && motion > 0.3 * prev_motion) return 1;
    return 0;
}

// ---- float32 filter of the float64 centroid search (pass_device.h: nn_point_filter) -----------------------------
// How far rounding the three coordinates of a point to float32 can move it, for coordinates up to `maxabs` in
// magnitude: half an ulp = 2^-24 relative per coordinate, sqrt(3) for the vector, 1 % on top.
PCR_HD static inline double gn_filter_band(double maxabs) {
    return 1.7321 * 1.01 * maxabs * 5.9604644775390625e-8 + 1e-30;
}
// Bound (squared) and tracking margin of the float32 search that goes with a float64 search bound: the bound grows by
// the band (a rounded centroid beyond it is a true centroid beyond the float64 bound); the margin must leave, after the
// 0.99999 safety factor on the reported lower bound (<= 1e-5 of the bound) and the band on either side, a positive gap:
// lbq = (sqrt(best) + mu) * 0.99999 - band  >  sqrt(best) + band  >=  the winner's float64 distance.
PCR_HD static inline void gn_filter_bounds(double band, double bound, float *bound2_ff, float *mu_ff) {
    const double bf = (bound + band) * 1.00002;
    *bound2_ff = (float)(bf * bf * 1.000001);
    *mu_ff = (float)(2.0 * band + 3e-5 * bf);
}
