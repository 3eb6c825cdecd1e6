/*
 * pcr_oracle.c -- CPU restatement of the reference's per-iteration registration path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it; the product path
 * (point_cloud_registration_amd/) never imports, links or calls anything in oracle/.
 *
 * Parity status: PINNED.  Every function below is checked by tests/test_oracle_golden.py
 * against golden vectors produced by importing the reference itself
 * (tests/golden/make_golden.py; scomup/point-cloud-registration @ 2025-05-09 with a
 * scipy-backed stand-in for its third-party pykdtree dependency, which is not
 * installable here -- exact 1-NN has a unique answer up to exact ties).
 *
 * Each function cites the reference file:line it restates (paths relative to
 * /root/reference/point_cloud_registration/).  Nothing is copied: the reference is
 * NumPy; this is scalar C with the same arithmetic definitions.
 *
 * Arithmetic conventions shared with the HIP kernels (so oracle-vs-kernel parity is
 * tight, ~1e-12, and only summation order differs):
 *   - transform: T cast to float32, x' = ((R00*x + R01*y) + R02*z) + t0 in float32,
 *     no FMA contraction (build with -ffp-contract=off);
 *   - point NN: d2 = fma(dz, dz, fma(dy, dy, dx*dx)) in float32 (fmaf is exactly specified, so the
 *     CPU and the GPU agree bit for bit), dist = sqrtf(d2), ties broken by the smaller target
 *     index;  centroid NN: d2 = (dx*dx + dy*dy) + dz*dz in float64;
 *   - per-point residuals/Jacobians in float64 from the float32 inputs, all sums float64
 *     (the reference accumulates a few blocks in float32, quirk Q5: the oracle is the
 *     more accurate of the two and agrees with the reference to ~1e-6 of max|H|).
 */

#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))

enum { ORC_ICP = 0, ORC_PLANE = 1, ORC_VPLANE = 2, ORC_NDT = 3 };
enum { ORC_FLAG_ICP_RR_QUIRK = 1,      /* quirk Q1: g1 = sum p x (R r)  (icp.py:53-54) */
       ORC_FLAG_GATE_F64 = 2 };        /* quirk Q6: a float64 target's tree returns float64 distances, so plane_icp.py:41
                                          gates in float64 (records stay the float32 copy, plane_icp.py:20,44) */

ORC_API int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
/* size of the OpenMP team of every later call (bench.py: the container's CPU quota, not the visible CPUs) */
ORC_API void orc_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ------------------------------------------------------------------ A2: transform
 * math_tools.py:111-113 transform_points, called with T.astype(float32)
 * (icp.py:32, plane_icp.py:39, voxelized_plane_icp.py:32, ndt.py:26).              */
ORC_API void orc_transform(const double T[16], const float *src, int64_t n, float *out) {
    const float r00 = (float)T[0], r01 = (float)T[1], r02 = (float)T[2], t0 = (float)T[3];
    const float r10 = (float)T[4], r11 = (float)T[5], r12 = (float)T[6], t1 = (float)T[7];
    const float r20 = (float)T[8], r21 = (float)T[9], r22 = (float)T[10], t2 = (float)T[11];
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        const float x = src[3 * i], y = src[3 * i + 1], z = src[3 * i + 2];
        out[3 * i] = ((r00 * x + r01 * y) + r02 * z) + t0;
        out[3 * i + 1] = ((r10 * x + r11 * y) + r12 * z) + t1;
        out[3 * i + 2] = ((r20 * x + r21 * y) + r22 * z) + t2;
    }
}

/* ------------------------------------------------------------------ A3: exact 1-NN
 * kdtree.py:18-21 KDTree(data).query(pts) -> (dist, idx); pykdtree itself is a
 * third-party dependency (setup.py:20) absent from /root/reference: restated here as
 * the mathematical definition (exhaustive search).                                   */
static inline float d2f(const float *a, const float *b) {
    const float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    return fmaf(dz, dz, fmaf(dy, dy, dx * dx));     /* two fused multiply-adds, one rounding each */
}
static inline double d2d(const float *q, const double *c) {
    const double dx = (double)q[0] - c[0], dy = (double)q[1] - c[1], dz = (double)q[2] - c[2];
    return (dx * dx + dy * dy) + dz * dz;
}

ORC_API void orc_nn_brute_f32(const float *tgt, int64_t nt, const float *q, int64_t m,
                              float *dist, int64_t *idx) {
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t i = 0; i < m; ++i) {
        float best = INFINITY;
        int64_t bi = -1;
        for (int64_t j = 0; j < nt; ++j) {
            const float d = d2f(q + 3 * i, tgt + 3 * j);
            if (d < best) { best = d; bi = j; }
        }
        dist[i] = sqrtf(best);
        idx[i] = bi;
    }
}

ORC_API void orc_nn_brute_f64(const double *tgt, int64_t nt, const float *q, int64_t m,
                              double *dist, int64_t *idx) {
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t i = 0; i < m; ++i) {
        double best = INFINITY;
        int64_t bi = -1;
        for (int64_t j = 0; j < nt; ++j) {
            const double d = d2d(q + 3 * i, tgt + 3 * j);
            if (d < best) { best = d; bi = j; }
        }
        dist[i] = sqrt(best);
        idx[i] = bi;
    }
}

/* k-NN by exhaustive search, neighbours sorted by (d2, idx) ascending
 * (kdtree.py:18-21 query(points, k) as used by estimate_normals.py:39).              */
ORC_API void orc_knn_brute_f32(const float *tgt, int64_t nt, const float *q, int64_t m, int k,
                               float *dist, int64_t *idx) {
#pragma omp parallel for schedule(dynamic, 16)
    for (int64_t i = 0; i < m; ++i) {
        float *bd = dist + (int64_t)k * i;
        int64_t *bi = idx + (int64_t)k * i;
        int cnt = 0;
        for (int64_t j = 0; j < nt; ++j) {
            const float d = d2f(q + 3 * i, tgt + 3 * j);
            if (cnt == k && !(d < bd[k - 1])) continue;
            int p = cnt < k ? cnt : k - 1;
            while (p > 0 && bd[p - 1] > d) { bd[p] = bd[p - 1]; bi[p] = bi[p - 1]; --p; }
            bd[p] = d; bi[p] = j;
            if (cnt < k) ++cnt;
        }
        for (int p = 0; p < k; ++p) bd[p] = p < cnt ? sqrtf(bd[p]) : INFINITY;
        for (int p = cnt; p < k; ++p) bi[p] = nt;
    }
}

/* Grid-accelerated exact 1-NN (for the CPU baseline and for large-size parity runs where
 * the exhaustive search is too slow).  Same answer as orc_nn_brute_* (tested), including
 * the tie rule.  Classic ring expansion over a dense cell grid; r_max bounds the search:
 * queries with no target point closer than r_max return idx = -1, dist = INFINITY.      */
typedef struct {
    int is_f64;
    int64_t n;
    double org[3], cell, inv;
    int64_t dim[3];
    int64_t *start;    /* dim0*dim1*dim2 + 1 */
    int64_t *order;    /* point index, cell-sorted, ascending index inside a cell */
    const void *pts;
} orc_grid;

static inline void grid_get(const orc_grid *g, int64_t j, double p[3]) {
    if (g->is_f64) { const double *s = (const double *)g->pts + 3 * j; p[0] = s[0]; p[1] = s[1]; p[2] = s[2]; }
    else { const float *s = (const float *)g->pts + 3 * j; p[0] = s[0]; p[1] = s[1]; p[2] = s[2]; }
}
static inline int64_t clampi(int64_t v, int64_t lo, int64_t hi) { return v < lo ? lo : (v > hi ? hi : v); }

ORC_API orc_grid *orc_grid_build(const void *pts, int is_f64, int64_t n, double cell) {
    orc_grid *g = (orc_grid *)calloc(1, sizeof(orc_grid));
    g->is_f64 = is_f64; g->n = n; g->pts = pts;
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int64_t j = 0; j < n; ++j) {
        double p[3]; grid_get(g, j, p);
        for (int a = 0; a < 3; ++a) { if (p[a] < lo[a]) lo[a] = p[a]; if (p[a] > hi[a]) hi[a] = p[a]; }
    }
    if (n == 0) { for (int a = 0; a < 3; ++a) { lo[a] = 0; hi[a] = 0; } }
    /* coarsen until the dense grid is affordable */
    for (;;) {
        double cells = 1;
        for (int a = 0; a < 3; ++a) { g->dim[a] = (int64_t)floor((hi[a] - lo[a]) / cell) + 1; cells *= (double)g->dim[a]; }
        if (cells <= 4.0e8) break;
        cell *= 2;
    }
    g->cell = cell; g->inv = 1.0 / cell;
    for (int a = 0; a < 3; ++a) g->org[a] = lo[a];
    const int64_t nc = g->dim[0] * g->dim[1] * g->dim[2];
    g->start = (int64_t *)calloc((size_t)nc + 1, sizeof(int64_t));
    g->order = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    int64_t *cid = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    for (int64_t j = 0; j < n; ++j) {
        double p[3]; grid_get(g, j, p);
        int64_t c[3];
        for (int a = 0; a < 3; ++a) c[a] = clampi((int64_t)floor((p[a] - g->org[a]) * g->inv), 0, g->dim[a] - 1);
        cid[j] = (c[2] * g->dim[1] + c[1]) * g->dim[0] + c[0];
        g->start[cid[j] + 1]++;
    }
    for (int64_t c = 0; c < nc; ++c) g->start[c + 1] += g->start[c];
    int64_t *fill = (int64_t *)malloc(sizeof(int64_t) * (size_t)nc);
    memcpy(fill, g->start, sizeof(int64_t) * (size_t)nc);
    for (int64_t j = 0; j < n; ++j) g->order[fill[cid[j]]++] = j;   /* ascending j inside a cell */
    free(fill); free(cid);
    return g;
}

ORC_API void orc_grid_free(orc_grid *g) {
    if (!g) return;
    free(g->start); free(g->order); free(g);
}

/* one query; the candidate distance uses the exact same float32 / float64 expression as
 * the exhaustive search so the two agree bit for bit */
static void grid_nn_one(const orc_grid *g, const float *q, double r_max, double *dist_out, int64_t *idx_out) {
    double best = INFINITY;            /* squared, in the arithmetic of the data type */
    int64_t bi = -1;
    double qd[3] = {q[0], q[1], q[2]};
    int64_t c[3];
    const double slack = 1e-9 * (g->cell + fabs(qd[0]) + fabs(qd[1]) + fabs(qd[2]));
    for (int a = 0; a < 3; ++a) c[a] = (int64_t)floor((qd[a] - g->org[a]) * g->inv);
    /* distance from the query to the grid's box: rings inside it cannot hold points */
    int64_t kmax = 0;
    for (int a = 0; a < 3; ++a) {
        int64_t k1 = llabs(c[a] - 0), k2 = llabs(c[a] - (g->dim[a] - 1));
        int64_t km = k1 > k2 ? k1 : k2;
        if (km > kmax) kmax = km;
    }
    const int64_t kcap = isfinite(r_max) ? (int64_t)ceil(r_max * g->inv) + 1 : kmax;
    if (kcap < kmax) kmax = kcap;
    for (int64_t k = 0; k <= kmax; ++k) {
        /* everything at Chebyshev cell distance >= k is at least (k-1)*cell away (+frac) */
        if (k >= 1) {
            double lb = (double)(k - 1) * g->cell - slack;
            if (lb > 0) {
                const double lim = isfinite(r_max) && r_max * r_max < best ? r_max * r_max : best;
                if (lb * lb > lim) break;
            }
        }
        for (int64_t dz = -k; dz <= k; ++dz) {
            const int64_t z = c[2] + dz;
            if (z < 0 || z >= g->dim[2]) continue;
            for (int64_t dy = -k; dy <= k; ++dy) {
                const int64_t y = c[1] + dy;
                if (y < 0 || y >= g->dim[1]) continue;
                const int on_shell = (llabs(dz) == k) || (llabs(dy) == k);
                const int64_t step = (on_shell || k == 0) ? 1 : 2 * k;   /* interior rows: only the two end cells */
                for (int64_t dx = -k; dx <= k; dx += step) {
                    const int64_t x = c[0] + dx;
                    if (x < 0 || x >= g->dim[0]) continue;
                    const int64_t cell_id = (z * g->dim[1] + y) * g->dim[0] + x;
                    for (int64_t s = g->start[cell_id]; s < g->start[cell_id + 1]; ++s) {
                        const int64_t j = g->order[s];
                        double d;
                        if (g->is_f64) d = d2d(q, (const double *)g->pts + 3 * j);
                        else d = (double)d2f(q, (const float *)g->pts + 3 * j);
                        if (d < best || (d == best && j < bi)) { best = d; bi = j; }
                    }
                }
            }
        }
    }
    double dist = g->is_f64 ? sqrt(best) : (double)sqrtf((float)best);
    if (bi >= 0 && isfinite(r_max) && !(dist < r_max)) { bi = -1; dist = INFINITY; }
    if (bi < 0) dist = INFINITY;
    *dist_out = dist; *idx_out = bi;
}

/* dist is written as float64 in both modes (the float32 mode's values are exactly
 * representable); r_max = INFINITY for an unbounded search */
ORC_API void orc_grid_nn(const orc_grid *g, const float *q, int64_t m, double r_max,
                         double *dist, int64_t *idx) {
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t i = 0; i < m; ++i) grid_nn_one(g, q + 3 * i, r_max, dist + i, idx + i);
}

/* ------------------------------------------------------------------ A4-A9: linearize
 * out[29] = 21 upper-triangle entries of H (row-major: 00 01 .. 05 11 12 .. 55),
 * g[6], e2, correspondence count.                                                      */
typedef struct { double v[29]; } acc29;

static inline void acc_outer6(double *h, const double J[6], double w) {
    int p = 0;
    for (int a = 0; a < 6; ++a)
        for (int b = a; b < 6; ++b) h[p++] += w * J[a] * J[b];
}

/* icp.py:24-57 (vectorised) with the closed-form H from moments; quirk Q1 honoured when
 * ORC_FLAG_ICP_RR_QUIRK is set (g1 = sum p x (R r)), otherwise the consistent J^T r of the
 * loop version icp.py:59-90 (g1 = sum p x (R^T r)).                                      */
static void lin_icp(const double T[16], const float *src, const float *st, int64_t n,
                    const float *tgt, const double *dist, const int64_t *idx, double max_dist,
                    unsigned flags, double out[29]) {
    const double R[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]};
    double cnt = 0, sp[3] = {0, 0, 0}, mm[6] = {0, 0, 0, 0, 0, 0}, sr[3] = {0, 0, 0}, sg[3] = {0, 0, 0}, e2 = 0;
    const float md = (float)max_dist;
#pragma omp parallel
    {
        double lc = 0, lsp[3] = {0, 0, 0}, lmm[6] = {0, 0, 0, 0, 0, 0}, lsr[3] = {0, 0, 0}, lsg[3] = {0, 0, 0}, le2 = 0;
#pragma omp for schedule(static)
        for (int64_t i = 0; i < n; ++i) {
            if (idx[i] < 0 || !((float)dist[i] < md)) continue;          /* icp.py:34 strict gate */
            const float *q = tgt + 3 * idx[i];
            const double r[3] = {(double)(st[3 * i] - q[0]), (double)(st[3 * i + 1] - q[1]), (double)(st[3 * i + 2] - q[2])};
            const double x = src[3 * i], y = src[3 * i + 1], z = src[3 * i + 2];
            lc += 1; lsp[0] += x; lsp[1] += y; lsp[2] += z;
            lmm[0] += x * x; lmm[1] += x * y; lmm[2] += x * z; lmm[3] += y * y; lmm[4] += y * z; lmm[5] += z * z;
            lsr[0] += r[0]; lsr[1] += r[1]; lsr[2] += r[2];
            double v[3];
            if (flags & ORC_FLAG_ICP_RR_QUIRK) {
                for (int a = 0; a < 3; ++a) v[a] = R[3 * a] * r[0] + R[3 * a + 1] * r[1] + R[3 * a + 2] * r[2];
            } else {
                for (int a = 0; a < 3; ++a) v[a] = R[a] * r[0] + R[3 + a] * r[1] + R[6 + a] * r[2];
            }
            lsg[0] += y * v[2] - z * v[1]; lsg[1] += z * v[0] - x * v[2]; lsg[2] += x * v[1] - y * v[0];
            le2 += r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
        }
#pragma omp critical
        {
            cnt += lc; e2 += le2;
            for (int a = 0; a < 3; ++a) { sp[a] += lsp[a]; sr[a] += lsr[a]; sg[a] += lsg[a]; }
            for (int a = 0; a < 6; ++a) mm[a] += lmm[a];
        }
    }
    /* H_ll = M I (icp.py:43); H_lr = -R skew(sum p) (icp.py:44); H_rr from moments (math_tools.py:44-58) */
    double H[6][6]; memset(H, 0, sizeof H);
    for (int a = 0; a < 3; ++a) H[a][a] = cnt;
    const double S[9] = {0, -sp[2], sp[1], sp[2], 0, -sp[0], -sp[1], sp[0], 0};
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) {
            double s = 0;
            for (int c = 0; c < 3; ++c) s += R[3 * a + c] * S[3 * c + b];
            H[a][3 + b] = -s;
        }
    H[3][3] = mm[3] + mm[5]; H[3][4] = -mm[1]; H[3][5] = -mm[2];
    H[4][4] = mm[0] + mm[5]; H[4][5] = -mm[4]; H[5][5] = mm[0] + mm[3];
    int p = 0;
    for (int a = 0; a < 6; ++a) for (int b = a; b < 6; ++b) out[p++] = H[a][b];
    for (int a = 0; a < 3; ++a) { out[21 + a] = sr[a]; out[24 + a] = sg[a]; }
    out[27] = e2; out[28] = cnt;
}

/* plane_icp.py:30-69 (target point + per-point normal, float32 records) and
 * voxelized_plane_icp.py:23-64 (voxel mean + voxel normal, float64 records).           */
static void lin_plane(const double T[16], const float *src, const float *st, int64_t n,
                      const void *q_rec, const void *n_rec, int rec_f64,
                      const double *dist, const int64_t *idx, double max_dist, int gate_f64, double out[29]) {
    const double R[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]};
    memset(out, 0, 29 * sizeof(double));
#pragma omp parallel
    {
        acc29 l; memset(&l, 0, sizeof l);
#pragma omp for schedule(static)
        for (int64_t i = 0; i < n; ++i) {
            if (idx[i] < 0) continue;
            if (rec_f64 || gate_f64) { if (!(dist[i] < max_dist)) continue; }
            else { if (!((float)dist[i] < (float)max_dist)) continue; }   /* plane_icp.py:41 */
            double nv[3], diff[3];
            if (rec_f64) {
                const double *q = (const double *)q_rec + 3 * idx[i], *nn = (const double *)n_rec + 3 * idx[i];
                for (int a = 0; a < 3; ++a) { nv[a] = nn[a]; diff[a] = (double)st[3 * i + a] - q[a]; }
            } else {
                const float *q = (const float *)q_rec + 3 * idx[i], *nn = (const float *)n_rec + 3 * idx[i];
                for (int a = 0; a < 3; ++a) { nv[a] = nn[a]; diff[a] = (double)(st[3 * i + a] - q[a]); }
            }
            const double r = (nv[0] * diff[0] + nv[1] * diff[1]) + nv[2] * diff[2];       /* plane_icp.py:49 */
            const double a_ = R[0] * nv[0] + R[3] * nv[1] + R[6] * nv[2];                  /* R^T n, plane_icp.py:51 */
            const double b_ = R[1] * nv[0] + R[4] * nv[1] + R[7] * nv[2];
            const double c_ = R[2] * nv[0] + R[5] * nv[1] + R[8] * nv[2];
            const double x = src[3 * i], y = src[3 * i + 1], z = src[3 * i + 2];
            const double J[6] = {nv[0], nv[1], nv[2],
                                 -z * b_ + y * c_, z * a_ - x * c_, -y * a_ + x * b_};    /* math_tools.py:22-31 */
            acc_outer6(l.v, J, 1.0);
            for (int a = 0; a < 6; ++a) l.v[21 + a] += J[a] * r;
            l.v[27] += r * r; l.v[28] += 1;
        }
#pragma omp critical
        for (int a = 0; a < 29; ++a) out[a] += l.v[a];
    }
}

/* ndt.py:24-57: d = Rp + t - mu, C = inverse covariance (6 unique: xx xy xz yy yz zz),
 * J = [I, -R skew(p)].                                                                   */
static void lin_ndt(const double T[16], const float *src, const float *st, int64_t n,
                    const double *mean, const double *icov6, const double *dist, const int64_t *idx,
                    double max_dist, double out[29]) {
    const double R[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]};
    memset(out, 0, 29 * sizeof(double));
#pragma omp parallel
    {
        acc29 l; memset(&l, 0, sizeof l);
#pragma omp for schedule(static)
        for (int64_t i = 0; i < n; ++i) {
            if (idx[i] < 0 || !(dist[i] < max_dist)) continue;                             /* ndt.py:32 */
            const double *mu = mean + 3 * idx[i], *c6 = icov6 + 6 * idx[i];
            const double C[3][3] = {{c6[0], c6[1], c6[2]}, {c6[1], c6[3], c6[4]}, {c6[2], c6[4], c6[5]}};
            const double d[3] = {(double)st[3 * i] - mu[0], (double)st[3 * i + 1] - mu[1], (double)st[3 * i + 2] - mu[2]};
            const double x = src[3 * i], y = src[3 * i + 1], z = src[3 * i + 2];
            const double S[3][3] = {{0, -z, y}, {z, 0, -x}, {-y, x, 0}};
            double J[3][6];                                                                 /* ndt.py:40 J1 = -R skew(p) */
            for (int a = 0; a < 3; ++a) {
                for (int b = 0; b < 3; ++b) {
                    J[a][b] = a == b ? 1.0 : 0.0;
                    double s = 0;
                    for (int c = 0; c < 3; ++c) s += R[3 * a + c] * S[c][b];
                    J[a][3 + b] = -s;
                }
            }
            double CJ[3][6], Cd[3];
            for (int a = 0; a < 3; ++a) {
                Cd[a] = C[a][0] * d[0] + C[a][1] * d[1] + C[a][2] * d[2];
                for (int b = 0; b < 6; ++b) CJ[a][b] = C[a][0] * J[0][b] + C[a][1] * J[1][b] + C[a][2] * J[2][b];
            }
            int p = 0;
            for (int a = 0; a < 6; ++a)
                for (int b = a; b < 6; ++b)
                    l.v[p++] += J[0][a] * CJ[0][b] + J[1][a] * CJ[1][b] + J[2][a] * CJ[2][b];
            for (int a = 0; a < 6; ++a) l.v[21 + a] += J[0][a] * Cd[0] + J[1][a] * Cd[1] + J[2][a] * Cd[2];
            l.v[27] += d[0] * Cd[0] + d[1] * Cd[1] + d[2] * Cd[2];
            l.v[28] += 1;
        }
#pragma omp critical
        for (int a = 0; a < 29; ++a) out[a] += l.v[a];
    }
}

/* Reduce step given correspondences.  rec_a / rec_b by kind:
 *   ICP    rec_a = target xyz float32 (N_t,3), rec_b unused
 *   PLANE  rec_a = target xyz float32,          rec_b = normals float32 (N_t,3)
 *   VPLANE rec_a = voxel means float64 (N_v,3), rec_b = voxel normals float64 (N_v,3)
 *   NDT    rec_a = voxel means float64,         rec_b = icov float64 (N_v,6)
 * dist is float64 storage in every mode (float32 values are exactly representable).     */
ORC_API int orc_linearize(int kind, const double T[16], const float *src, const float *src_trans, int64_t n,
                          const void *rec_a, const void *rec_b, const double *dist, const int64_t *idx,
                          double max_dist, unsigned flags, double out[29]) {
    switch (kind) {
    case ORC_ICP: lin_icp(T, src, src_trans, n, (const float *)rec_a, dist, idx, max_dist, flags, out); return 0;
    case ORC_PLANE: lin_plane(T, src, src_trans, n, rec_a, rec_b, 0, dist, idx, max_dist, (flags & ORC_FLAG_GATE_F64) != 0, out); return 0;
    case ORC_VPLANE: lin_plane(T, src, src_trans, n, rec_a, rec_b, 1, dist, idx, max_dist, 1, out); return 0;
    case ORC_NDT: lin_ndt(T, src, src_trans, n, (const double *)rec_a, (const double *)rec_b, dist, idx, max_dist, out); return 0;
    default: return -1;
    }
}

/* ------------------------------------------------------------------ T1-T3: voxel build */
static inline int64_t pymod(int64_t a, int64_t m) { int64_t r = a % m; return r < 0 ? r + m : r; }

/* voxel.py:12-21 get_keys.  floor(points / voxel_size) is evaluated in the dtype of the
 * points (float32 clouds divide in float32), then cast to int64.                        */
ORC_API void orc_voxel_keys(const void *pts, int is_f64, int64_t n, double voxel_size, int64_t *keys) {
    const int64_t P = 116101, M = 10000000000LL;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        int64_t v[3];
        for (int a = 0; a < 3; ++a) {
            if (is_f64) v[a] = (int64_t)floor(((const double *)pts)[3 * i + a] / voxel_size);
            else v[a] = (int64_t)floorf(((const float *)pts)[3 * i + a] / (float)voxel_size);
        }
        keys[i] = pymod((pymod(v[2] * P, M) + v[1]) * P, M) + v[0];
    }
}

typedef struct { int64_t key, idx; } kv;
static int kv_cmp(const void *a, const void *b) {
    const kv *x = (const kv *)a, *y = (const kv *)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0);
}

/* symmetric 3x3 eigen-decomposition, cyclic Jacobi in float64; eigenvalues ascending,
 * evec[c] (3 doubles) the unit eigenvector of eval[c].  Stands in for numpy.linalg.eigh
 * (LAPACK) at voxel.py:157 and estimate_normals.py:73; eigenvector sign is arbitrary.    */
ORC_API void orc_eigh3(const double A[9], double eval[3], double evec[9]) {
    double a[3][3] = {{A[0], A[1], A[2]}, {A[1], A[4], A[5]}, {A[2], A[5], A[8]}};
    double v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 60; ++sweep) {
        const double off = fabs(a[0][1]) + fabs(a[0][2]) + fabs(a[1][2]);
        const double dia = fabs(a[0][0]) + fabs(a[1][1]) + fabs(a[2][2]);
        if (off <= 1e-300 || off <= 1e-18 * dia) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (a[p][q] == 0.0) continue;
                const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; ++k) {
                    const double akp = a[k][p], akq = a[k][q];
                    a[k][p] = c * akp - s * akq; a[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; ++k) {
                    const double apk = a[p][k], aqk = a[q][k];
                    a[p][k] = c * apk - s * aqk; a[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; ++k) {
                    const double vkp = v[k][p], vkq = v[k][q];
                    v[k][p] = c * vkp - s * vkq; v[k][q] = s * vkp + c * vkq;
                }
            }
    }
    int ord[3] = {0, 1, 2};
    for (int i = 0; i < 2; ++i)
        for (int j = i + 1; j < 3; ++j)
            if (a[ord[j]][ord[j]] < a[ord[i]][ord[i]]) { int t = ord[i]; ord[i] = ord[j]; ord[j] = t; }
    for (int c = 0; c < 3; ++c) {
        eval[c] = a[ord[c]][ord[c]];
        double nrm = 0;
        for (int k = 0; k < 3; ++k) nrm += v[k][ord[c]] * v[k][ord[c]];
        nrm = sqrt(nrm);
        for (int k = 0; k < 3; ++k) evec[3 * c + k] = v[k][ord[c]] / nrm;
    }
}

/* voxel.py:69-102 calc_icov: adjugate / determinant, det == 0 exactly -> 1e6.           */
ORC_API void orc_calc_icov(const double *cov9, int64_t nv, double *icov9) {
    for (int64_t i = 0; i < nv; ++i) {
        const double *m = cov9 + 9 * i;
        const double a = m[0], b = m[4], c = m[8], d = m[1], e = m[2], f = m[5];
        const double f2 = f * f, d2 = d * d, e2 = e * e;
        const double bc = b * c, ac = a * c, ab = a * b;
        const double dc = d * c, de = d * e, ef = e * f;
        const double af = a * f, df = d * f, eb = e * b;
        double det = a * bc + 2 * de * f - a * f2 - b * e2 - c * d2;
        if (det == 0) det = 1000000;
        const double c0 = (bc - f2) / det, c1 = -(dc - ef) / det, c2 = (df - eb) / det;
        const double c3 = (ac - e2) / det, c4 = -(af - de) / det, c5 = (ab - d2) / det;
        double *o = icov9 + 9 * i;
        o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c1; o[4] = c3; o[5] = c4; o[6] = c2; o[7] = c4; o[8] = c5;
    }
}

/* voxel.py:104-165 VoxelGrid.set_points.  Two calls: with mean == NULL only the counts
 * are returned (n_unique, n_kept) so the caller can size the outputs.  Voxel order is
 * ascending key (np.unique); per-voxel sums run in ascending point index (np.bincount). */
ORC_API int orc_voxel_build(const void *pts, int is_f64, int64_t n, double voxel_size, int min_points,
                            int64_t *n_unique, int64_t *n_kept,
                            double *mean, double *cov9, double *norm, int64_t *counts_kept, int64_t *keys_kept) {
    kv *s = (kv *)malloc(sizeof(kv) * (size_t)(n > 0 ? n : 1));
    int64_t *keys = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    orc_voxel_keys(pts, is_f64, n, voxel_size, keys);
    for (int64_t i = 0; i < n; ++i) { s[i].key = keys[i]; s[i].idx = i; }
    free(keys);
    qsort(s, (size_t)n, sizeof(kv), kv_cmp);
    int64_t nu = 0, nk = 0;
    for (int64_t b = 0; b < n;) {
        int64_t e = b;
        while (e < n && s[e].key == s[b].key) ++e;
        const int64_t cnt = e - b;
        ++nu;
        if (cnt >= min_points) {
            if (mean) {
                double sum[3] = {0, 0, 0};
                for (int64_t t = b; t < e; ++t) {
                    double p[3];
                    if (is_f64) { const double *q = (const double *)pts + 3 * s[t].idx; p[0] = q[0]; p[1] = q[1]; p[2] = q[2]; }
                    else { const float *q = (const float *)pts + 3 * s[t].idx; p[0] = q[0]; p[1] = q[1]; p[2] = q[2]; }
                    sum[0] += p[0]; sum[1] += p[1]; sum[2] += p[2];
                }
                double mu[3] = {sum[0] / (double)cnt, sum[1] / (double)cnt, sum[2] / (double)cnt};   /* voxel.py:118-121 */
                double cc[6] = {0, 0, 0, 0, 0, 0};
                for (int64_t t = b; t < e; ++t) {
                    double p[3];
                    if (is_f64) { const double *q = (const double *)pts + 3 * s[t].idx; p[0] = q[0]; p[1] = q[1]; p[2] = q[2]; }
                    else { const float *q = (const float *)pts + 3 * s[t].idx; p[0] = q[0]; p[1] = q[1]; p[2] = q[2]; }
                    const double dx = p[0] - mu[0], dy = p[1] - mu[1], dz = p[2] - mu[2];         /* voxel.py:125 */
                    cc[0] += dx * dx; cc[1] += dx * dy; cc[2] += dx * dz; cc[3] += dy * dy; cc[4] += dy * dz; cc[5] += dz * dz;
                }
                const double den = (double)(cnt - 1 > 1 ? cnt - 1 : 1);                            /* voxel.py:136 */
                for (int a = 0; a < 6; ++a) cc[a] /= den;
                double *m9 = cov9 + 9 * nk;
                m9[0] = cc[0]; m9[1] = cc[1]; m9[2] = cc[2]; m9[3] = cc[1]; m9[4] = cc[3]; m9[5] = cc[4];
                m9[6] = cc[2]; m9[7] = cc[4]; m9[8] = cc[5];
                for (int a = 0; a < 3; ++a) mean[3 * nk + a] = mu[a];
                double ev[3], evec[9];
                orc_eigh3(m9, ev, evec);
                for (int a = 0; a < 3; ++a) norm[3 * nk + a] = evec[a];                            /* voxel.py:157-158 */
                if (counts_kept) counts_kept[nk] = cnt;
                if (keys_kept) keys_kept[nk] = s[b].key;
            }
            ++nk;
        }
        b = e;
    }
    free(s);
    *n_unique = nu; *n_kept = nk;
    return 0;
}

/* ------------------------------------------------------------------ N2: k-NN PCA normals
 * estimate_normals.py:27-87 estimate_norm_with_tree.  The reference accumulates sum p and
 * sum p p^T over the k neighbours in FLOAT32, neighbour by neighbour (:56-65), forms
 * cov = E[pp^T] - mu mu^T in float32 (:71-72) and takes eigh's first eigenvector (:75-76).
 * compat != 0 reproduces that float32 single-pass arithmetic (then promotes the 3x3 to
 * float64 for the eigen-solve); compat == 0 is a centred float64 covariance.            */
ORC_API void orc_normals_from_knn(const float *pts, int64_t n, const int64_t *knn_idx, int k, int compat,
                                  float *normals) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        double cov[9];
        const int64_t *nb = knn_idx + (int64_t)k * i;
        if (compat) {
            float sum[3] = {0, 0, 0}, pp[6] = {0, 0, 0, 0, 0, 0};
            for (int j = 0; j < k; ++j) {
                const float *p = pts + 3 * nb[j];
                const float x = p[0], y = p[1], z = p[2];
                sum[0] += x; sum[1] += y; sum[2] += z;
                pp[0] += x * x; pp[1] += x * y; pp[2] += x * z; pp[3] += y * y; pp[4] += y * z; pp[5] += z * z;
            }
            const float kf = (float)k;
            const float mx = sum[0] / kf, my = sum[1] / kf, mz = sum[2] / kf;
            const float c00 = pp[0] / kf - mx * mx, c01 = pp[1] / kf - mx * my, c02 = pp[2] / kf - mx * mz;
            const float c11 = pp[3] / kf - my * my, c12 = pp[4] / kf - my * mz, c22 = pp[5] / kf - mz * mz;
            cov[0] = c00; cov[1] = c01; cov[2] = c02; cov[3] = c01; cov[4] = c11; cov[5] = c12; cov[6] = c02; cov[7] = c12; cov[8] = c22;
        } else {
            double mu[3] = {0, 0, 0};
            for (int j = 0; j < k; ++j) for (int a = 0; a < 3; ++a) mu[a] += pts[3 * nb[j] + a];
            for (int a = 0; a < 3; ++a) mu[a] /= k;
            double c[6] = {0, 0, 0, 0, 0, 0};
            for (int j = 0; j < k; ++j) {
                const double dx = pts[3 * nb[j]] - mu[0], dy = pts[3 * nb[j] + 1] - mu[1], dz = pts[3 * nb[j] + 2] - mu[2];
                c[0] += dx * dx; c[1] += dx * dy; c[2] += dx * dz; c[3] += dy * dy; c[4] += dy * dz; c[5] += dz * dz;
            }
            for (int a = 0; a < 6; ++a) c[a] /= k;
            cov[0] = c[0]; cov[1] = c[1]; cov[2] = c[2]; cov[3] = c[1]; cov[4] = c[3]; cov[5] = c[4]; cov[6] = c[2]; cov[7] = c[4]; cov[8] = c[5];
        }
        double ev[3], evec[9];
        orc_eigh3(cov, ev, evec);
        for (int a = 0; a < 3; ++a) normals[3 * i + a] = (float)evec[a];
    }
}

/* ------------------------------------------------------------------ A10: GN step pieces */
/* numpy.linalg.solve(H, g) (registration.py:103): LU with partial pivoting; returns 1 when
 * an exact zero pivot appears (LAPACK info > 0 -> LinAlgError "Singular matrix", quirk Q7). */
ORC_API int orc_solve6(const double H[36], const double g[6], double x[6]) {
    double A[6][7];
    for (int i = 0; i < 6; ++i) { for (int j = 0; j < 6; ++j) A[i][j] = H[6 * i + j]; A[i][6] = g[i]; }
    for (int c = 0; c < 6; ++c) {
        int piv = c;
        for (int r = c + 1; r < 6; ++r) if (fabs(A[r][c]) > fabs(A[piv][c])) piv = r;
        if (A[piv][c] == 0.0) return 1;
        if (piv != c) for (int j = 0; j < 7; ++j) { double t = A[c][j]; A[c][j] = A[piv][j]; A[piv][j] = t; }
        for (int r = c + 1; r < 6; ++r) {
            const double f = A[r][c] / A[c][c];
            for (int j = c; j < 7; ++j) A[r][j] -= f * A[c][j];
        }
    }
    for (int i = 5; i >= 0; --i) {
        double s = A[i][6];
        for (int j = i + 1; j < 6; ++j) s -= A[i][j] * x[j];
        x[i] = s / A[i][i];
    }
    return 0;
}

/* math_tools.py:80-98 expSO3 with the first-order branch at theta^2 <= 1e-5 (quirk Q3). */
ORC_API void orc_expSO3(const double w[3], double R[9]) {
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    const double W[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
    if (th2 <= 1e-5) {
        for (int i = 0; i < 9; ++i) R[i] = W[i];
        R[0] += 1; R[4] += 1; R[8] += 1;
        return;
    }
    const double th = sqrt(th2);
    double K[9], KK[9];
    for (int i = 0; i < 9; ++i) K[i] = W[i] / th;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += K[3 * i + k] * K[3 * k + j];
            KK[3 * i + j] = s;
        }
    const double sn = sin(th), omc = 1.0 - cos(th);
    for (int i = 0; i < 9; ++i) R[i] = sn * K[i] + omc * KK[i];
    R[0] += 1; R[4] += 1; R[8] += 1;
}

/* math_tools.py:101-108 plus: T <- T @ [expSO3(dx[3:]), dx[:3]; 0 1] (quirk Q2).        */
ORC_API void orc_plus(const double T[16], const double dx[6], double out[16]) {
    double dR[9]; orc_expSO3(dx + 3, dR);
    double D[16] = {dR[0], dR[1], dR[2], dx[0], dR[3], dR[4], dR[5], dx[1], dR[6], dR[7], dR[8], dx[2], 0, 0, 0, 1};
    double r[16];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double s = 0;
            for (int k = 0; k < 4; ++k) s += T[4 * i + k] * D[4 * k + j];
            r[4 * i + j] = s;
        }
    memcpy(out, r, sizeof r);
}
