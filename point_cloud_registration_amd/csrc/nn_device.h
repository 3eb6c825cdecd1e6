// Exact bounded nearest-neighbour search over a dense cell grid (device functions).
//
// Replaces KDTree.query (reference kdtree.py:18-21 -> pykdtree) and VoxelGrid.query's
// KD-tree over centroids (voxel.py:165,171-179).  One lane = one query.  Points are stored
// cell-sorted (cell id = (z*ny + y)*nx + x) as 16-byte (float4) or 32-byte (double4) records
// with the original index bit-cast into w, so a ROW of cells x0..x1 at fixed (y, z) is one
// contiguous range [cell_start[row+x0], cell_start[row+x1+1]) of HBM: a ring of the search is
// a handful of contiguous row segments, not 26 scattered cells.
//
// Search: Chebyshev rings k = 0, 1, ... around the query's cell.  After ring k-1 every
// unvisited point is at least (k-1)*h + (distance to the nearest face of the query's cell)
// away, which certifies the current best and ends the search.  Inside a ring, rows are
// skipped by their (y, z) slab distance and the x extent is clipped to the remaining budget
// sqrt(best - dyz^2).  All bounds are loosened by geom.slack so that float rounding in the
// cell assignment can never prune the true nearest neighbour.  Ties in distance are broken by
// the smaller original index (the oracle's rule), which makes the result independent of the
// storage order.
//
// Arithmetic: point targets d2 = fma(dz, dz, fma(dy, dy, dx*dx)) in float32 (explicit fused
// multiply-adds, exactly specified); centroids d2 = (dx*dx + dy*dy) + dz*dz in float64 with no
// contraction (the TU is compiled with -ffp-contract=off) -- identical to oracle/pcr_oracle.c
// d2f / d2d, so both pick the same neighbour bit for bit.
#pragma once

#include "pcr_internal.h"

#define PCR_NONE 0xffffffffu

__device__ __forceinline__ uint32_t pt_orig(const float4 &p) { return __float_as_uint(p.w); }
__device__ __forceinline__ uint32_t pt_orig(const double4 &p) { return (uint32_t)__double_as_longlong(p.w); }

template <typename Real> struct RealTraits;
template <> struct RealTraits<float> {
    __device__ static __forceinline__ float inf() { return __int_as_float(0x7f800000); }
    __device__ static __forceinline__ float sqrt_rn(float x) { return __builtin_sqrtf(x); }   // correctly rounded (hipcc default); __fsqrt_rn maps to the NATIVE sqrt
    __device__ static __forceinline__ float sqrt_fast(float x) { return __fsqrt_rn(x); }   // v_sqrt_f32, 1 ulp
    __device__ static __forceinline__ float floor_(float x) { return floorf(x); }
};
template <> struct RealTraits<double> {
    __device__ static __forceinline__ double inf() { return __longlong_as_double(0x7ff0000000000000LL); }
    __device__ static __forceinline__ double sqrt_rn(double x) { return __builtin_sqrt(x); }
    __device__ static __forceinline__ double sqrt_fast(double x) { return __builtin_sqrt(x); }
    __device__ static __forceinline__ double floor_(double x) { return floor(x); }
};

// float32 squared distance, the one definition shared with oracle/pcr_oracle.c (d2f):
// fma(dz, dz, fma(dy, dy, dx*dx)) -- two fused multiply-adds, each rounded once.
__device__ __forceinline__ float dist2_f32(float dx, float dy, float dz) {
    return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
}

// ---- tracking searches (certified reuse of a match by the NEXT pass, kernels.hip: k_certify) -------
// A TRACK search also returns a lower bound on the distance from the query to every target point
// OTHER than the winner.  It keeps `second` = the smallest squared distance seen on any candidate that
// is not the winner, and prunes with  prune = min(second, (sqrt(best) + mu)^2)  instead of `best`: every
// point it never looked at is then farther than sqrt(prune) at the time it was skipped, so
//     lb2^2 = min(second, smallest prune value ever used)
// bounds all the others from below.  mu (per lane) caps the price: the search looks at most mu beyond
// the match (mu = 0: exactly the plain search, and lb2 = the match distance, i.e. no margin).
template <typename Real>
struct NNTrack {
    Real second;      // min squared distance over examined candidates other than the current winner
    Real prune;       // current pruning bound (>= best)
    Real pmin;        // smallest pruning bound used so far
    Real two_mu, mu2; // 2 mu, mu^2
    // TRACK == 2 (the float32 filter of the centroid search settles two-way near-ties itself): the runner-up's identity,
    // and `third` = min squared distance over examined candidates other than the winner AND the runner-up; the search
    // then prunes with min(third, (sqrt(best) + mu)^2), so that min(third, pmin) bounds everybody but those two
    Real third;
    uint32_t sec_o;
};

template <typename Real>
__device__ __forceinline__ void nn_track_init(NNTrack<Real> &tk, Real bound2, Real mu) {
    tk.second = bound2; tk.prune = bound2; tk.pmin = bound2; tk.two_mu = mu + mu; tk.mu2 = mu * mu;
    tk.third = bound2; tk.sec_o = 0xffffffffu;
}

// The search tracks (squared distance, original index) -- the oracle's tie rule -- and, for the reduce
// kernel's gather, the index `j` of the winner in the array it was read from (the cell-sorted array, or
// an extended list whose entries are translated through Geom::j_h right after ring 0).
template <typename Real, typename PT, int TRACK = 0>
__device__ __forceinline__ void nn_test(const PT &p, uint32_t j, Real qx, Real qy, Real qz,
                                        Real &best, uint32_t &bj, uint32_t &borig, NNTrack<Real> *tk = nullptr) {
    const Real dx = qx - (Real)p.x, dy = qy - (Real)p.y, dz = qz - (Real)p.z;
    const Real d = (dx * dx + dy * dy) + dz * dz;
    const uint32_t o = pt_orig(p);
    // straight-line selects (bitwise, not short-circuit): no exec-mask juggling in the hot loop
    const bool take = (d < best) | ((d == best) & (o < borig));
    if (TRACK) {
        // the loser of this comparison is "some other point" -- unless the candidate IS the current winner
        // seen again (batches over-read into the next cell, halo copies are revisited by ring 1)
        const bool same = (d == best) & (o == borig);
        const Real other = take ? best : d;
        tk->second = same ? tk->second : fmin(tk->second, other);
    }
    best = take ? d : best; bj = take ? j : bj; borig = take ? o : borig;
}

// float32 specialisation: d >= 0, so the bit patterns of squared distances order like the values and
// (distance, original index) packs into ONE unsigned 64-bit key -- "closer, ties to the smaller
// index" becomes a single 64-bit compare instead of three compares and two mask operations.
template <int TRACK>
__device__ __forceinline__ void nn_test_f32(const float4 &p, uint32_t j, float qx, float qy, float qz,
                                            float &best, uint32_t &bj, uint32_t &borig, NNTrack<float> *tk) {
    const float dx = qx - p.x, dy = qy - p.y, dz = qz - p.z;
    const float d = dist2_f32(dx, dy, dz);
    const uint32_t o = pt_orig(p);
    const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | o;
    const unsigned long long cur = ((unsigned long long)__float_as_uint(best) << 32) | borig;
    const bool take = key < cur;
    if (TRACK == 1) {
        const bool same = key == cur;
        const float other = take ? best : d;
        tk->second = same ? tk->second : fminf(tk->second, other);
    }
    if (TRACK == 2) {
        // the loser of the comparison, unless the candidate is the current winner or the current runner-up seen again
        // (halo copies are revisited by ring 1, batches over-read into the next cell): those change nothing
        const unsigned long long sec = ((unsigned long long)__float_as_uint(tk->second) << 32) | tk->sec_o;
        const bool again = (key == cur) | (key == sec);
        const float other = again ? __int_as_float(0x7f800000) : (take ? best : d);
        const uint32_t other_o = take ? borig : o;
        const bool closer = other < tk->second;                        // (strict: an equal later one goes to `third`)
        tk->third = fminf(tk->third, fmaxf(tk->second, other));        // whoever drops out of the first two
        tk->sec_o = closer ? other_o : tk->sec_o;
        tk->second = fminf(tk->second, other);
    }
    best = take ? d : best; bj = take ? j : bj; borig = take ? o : borig;
}
template <>
__device__ __forceinline__ void nn_test<float, float4, 0>(const float4 &p, uint32_t j, float qx, float qy, float qz,
                                                              float &best, uint32_t &bj, uint32_t &borig, NNTrack<float> *tk) {
    nn_test_f32<0>(p, j, qx, qy, qz, best, bj, borig, tk);
}
template <>
__device__ __forceinline__ void nn_test<float, float4, 1>(const float4 &p, uint32_t j, float qx, float qy, float qz,
                                                          float &best, uint32_t &bj, uint32_t &borig, NNTrack<float> *tk) {
    nn_test_f32<1>(p, j, qx, qy, qz, best, bj, borig, tk);
}
template <>
__device__ __forceinline__ void nn_test<float, float4, 2>(const float4 &p, uint32_t j, float qx, float qy, float qz,
                                                          float &best, uint32_t &bj, uint32_t &borig, NNTrack<float> *tk) {
    nn_test_f32<2>(p, j, qx, qy, qz, best, bj, borig, tk);
}

// refresh the pruning bound of a tracking search after `best` / `second` may have changed
template <typename Real, int TRACK = 1>
__device__ __forceinline__ void nn_track_refresh(NNTrack<Real> &tk, Real best) {
    typedef RealTraits<Real> RT;
    // (sqrt(best) + mu)^2; any value >= best is a valid bound, so the fast sqrt is fine: what is recorded in
    // pmin is the value that was actually used
    const Real infl = best + (tk.two_mu * RT::sqrt_fast(best) + tk.mu2);
    const Real p = fmin(TRACK == 2 ? tk.third : tk.second, infl);
    tk.prune = p;
    tk.pmin = fmin(tk.pmin, p);
}

// The search is latency-bound (one L2 round trip per dependent load, ~500 cycles): candidates are
// fetched PCR_NN_BATCH at a time so that many loads are in flight per lane before the first compare.
// A batch may run up to PCR_NN_BATCH-1 records past the end of the range: those are real points of the following cells
// (testing an extra true candidate can only help), and the array carries PCR_PTS_PAD sentinel
// records at +inf behind its last point, which never win a comparison.
#define PCR_PTS_PAD 16
#ifndef PCR_NN_BATCH
#define PCR_NN_BATCH 4   // measured: 8 costs occupancy (101 VGPR) and is slower -- except for small scans, where a pass is a
#endif                   // chain of dependent round trips and a batch is one of them: the fused small-scan kernel uses
#ifndef PCR_NN_BATCH_SMALL   // PCR_NN_BATCH_SMALL (100 k-point ICP harness scan 42.3 -> 38.4 us per pass with 8)
#define PCR_NN_BATCH_SMALL 8
#endif
template <typename Real, typename PT, int TRACK = 0, int B = PCR_NN_BATCH>
__device__ __forceinline__ void nn_scan_range(const PT *__restrict__ pts, uint32_t s, uint32_t e,
                                              Real qx, Real qy, Real qz, Real &best, uint32_t &bj, uint32_t &borig,
                                              NNTrack<Real> *tk = nullptr) {
    for (uint32_t j = s; j < e; j += B) {
        const PT *__restrict__ b = pts + j;
        PT p[B];
#ifdef PCR_NN_ABLATE_LOADS
        // developer TIMING experiment (wrong results): ONE record of the batch is fetched, the others are made up from it, the
        // arithmetic stays -- what the search would cost if a batch were one load instruction instead of four
        p[0] = b[0];
#pragma unroll
        for (int u = 1; u < B; ++u) { p[u] = p[0]; p[u].x += (Real)(1.0e-3 * u); }
#else
#pragma unroll
        for (int u = 0; u < B; ++u) p[u] = b[u];
#endif
#pragma unroll
        for (int u = 0; u < B; ++u) nn_test<Real, PT, TRACK>(p[u], j + u, qx, qy, qz, best, bj, borig, tk);
    }
    if (TRACK) nn_track_refresh<Real, TRACK>(*tk, best);
}

// Per-lane work counters, compiled in only for pcr_nn_counters (STATS = true).
struct NNStats { uint32_t rings, rows_loaded, rows_pruned, cand; };

// ---- ranges with leaf / group boxes (Geom::lbox; round 6) ---------------------------------------------------------------
// Squared distance from the query to a box of records, by the candidates' own formula: for every record p inside the box
// |q - p| >= the clamped delta per axis, float subtraction and dist2_f32 are monotone under round-to-nearest, so this is a lower
// bound on the COMPUTED distance of every record inside -- exactly, no slack.  A box is skipped when it exceeds `best`
// strictly: a record AT the best distance (which could win on the smaller index) is always looked at.
// A box is ONE 16-byte record (round 6, second form: two float4 per box made a converged query in a 600-point cell pay 90 loads,
// 2/3 of them boxes): {lo.x, lo.y, lo.z, extents}, the min corner exactly and the three extents as 10-bit counts of
// qe = cell edge / 256, rounded UP plus one (1023 = unbounded on that axis: a block that straddles rows of cells).  The decoded
// max corner fma(count, qe, lo) errs by an ulp of the coordinate against a margin of one qe: it never lies below a record.
__device__ __forceinline__ float box_d2(const float4 &b, float qe, float qx, float qy, float qz) {
    const uint32_t w = __float_as_uint(b.w);
    const uint32_t ex = w & 1023u, ey = (w >> 10) & 1023u, ez = (w >> 20) & 1023u;
    const float inf = __int_as_float(0x7f800000);
    const float hx = ex == 1023u ? inf : __builtin_fmaf((float)ex, qe, b.x);
    const float hy = ey == 1023u ? inf : __builtin_fmaf((float)ey, qe, b.y);
    const float hz = ez == 1023u ? inf : __builtin_fmaf((float)ez, qe, b.z);
    const float dx = fmaxf(fmaxf(b.x - qx, qx - hx), 0.f);
    const float dy = fmaxf(fmaxf(b.y - qy, qy - hy), 0.f);
    const float dz = fmaxf(fmaxf(b.z - qz, qz - hz), 0.f);
    return dist2_f32(dx, dy, dz);
}
// Records [s, e) of `pts`: short ranges plainly; longer ones group by group (64 records) and leaf by leaf (8 records), each
// behind its box.  A heavy cell of a LiDAR sweep (hundreds to thousands of records on one ring line) costs a few dozen box
// tests and two or three leaves instead of every record.
#ifndef PCR_LB_BATCHED
#define PCR_LB_BATCHED 1
#endif
template <int B = PCR_NN_BATCH, bool STATS = false>
__device__ __forceinline__ void nn_scan_range_lb(const PtF *__restrict__ pts, const float4 *__restrict__ lbox, const float4 *__restrict__ gbox, float qe,
                                                 uint32_t s, uint32_t e, float qx, float qy, float qz,
                                                 float &best, uint32_t &bj, uint32_t &borig, NNStats *st = nullptr) {
    if (e - s <= PCR_LB_MIN) {
        if (STATS) st->cand += ((e - s + B - 1) / B) * B;
        nn_scan_range<float, PtF, 0, B>(pts, s, e, qx, qy, qz, best, bj, borig, nullptr);
        return;
    }
    const uint32_t g1 = (e - 1u) >> 6;
#if PCR_LB_BATCHED
    // Boxes are requested FOUR at a time (8 loads in flight) instead of one by one, and -- what matters -- the NEAREST box of a
    // batch is opened first: a range is entered with `best` still at the search bound (the gate, 2 m), against which every box
    // of the cell passes; the first version opened them in storage order and paid ~130 loads per converged query in a
    // 600-point cell before its bound became useful.  With the nearest group and, inside it, the nearest leaf first, the bound
    // is at the match distance after one leaf and everything else is re-tested against THAT.  Survivors are kept in bit masks
    // (no dynamically indexed register arrays); the two phases share one instance of the loop body.
    const uint32_t l_first = s >> 3, l_last = (e - 1u) >> 3;
    const float inf = __int_as_float(0x7f800000);
    for (uint32_t G = s >> 6; G <= g1; G += 4u) {
        float dg[4];
#pragma unroll
        for (uint32_t u = 0; u < 4u; ++u) {
            const uint32_t gi = min(G + u, g1);
            const float4 bx = gbox[gi];
            dg[u] = G + u <= g1 ? box_d2(bx, qe, qx, qy, qz) : inf;
        }
        if (STATS) st->cand += 4;
        uint32_t gmin = 0; float dgm = dg[0];
#pragma unroll
        for (uint32_t u = 1; u < 4u; ++u) { const bool c = dg[u] < dgm; gmin = c ? u : gmin; dgm = c ? dg[u] : dgm; }
#pragma unroll 1
        for (int gphase = 0; gphase < 2; ++gphase) {
            uint32_t gm = 0;
            if (gphase == 0) gm = dgm <= best ? 1u << gmin : 0u;
            else {
#pragma unroll
                for (uint32_t u = 0; u < 4u; ++u) gm |= (u != gmin && dg[u] <= best ? 1u : 0u) << u;
            }
            while (gm) {
                const uint32_t Gi = G + (uint32_t)__builtin_ctz(gm);
                gm &= gm - 1u;
                const uint32_t l0 = max(l_first, Gi << 3), l1 = min(l_last, (Gi << 3) + 7u);
                float dl[8];
#pragma unroll
                for (uint32_t v = 0; v < 8u; ++v) {
                    const uint32_t li = min(l0 + v, l1);
                    const float4 bx = lbox[li];
                    dl[v] = l0 + v <= l1 ? box_d2(bx, qe, qx, qy, qz) : inf;
                }
                if (STATS) st->cand += 8;
                uint32_t lmin = 0; float dlm = dl[0];
#pragma unroll
                for (uint32_t v = 1; v < 8u; ++v) { const bool c = dl[v] < dlm; lmin = c ? v : lmin; dlm = c ? dl[v] : dlm; }
#pragma unroll 1
                for (int lphase = 0; lphase < 2; ++lphase) {
                    uint32_t lm = 0;
                    if (lphase == 0) lm = dlm <= best ? 1u << lmin : 0u;
                    else {
#pragma unroll
                        for (uint32_t v = 0; v < 8u; ++v) lm |= (v != lmin && dl[v] <= best ? 1u : 0u) << v;
                    }
                    while (lm) {
                        const uint32_t Li = l0 + (uint32_t)__builtin_ctz(lm);
                        lm &= lm - 1u;
                        const uint32_t a = max(s, Li << 3), b = min(e, (Li << 3) + 8u);
                        if (STATS) st->cand += 8;
                        nn_scan_range<float, PtF, 0, 8>(pts, a, b, qx, qy, qz, best, bj, borig, nullptr);
                    }
                }
            }
        }
    }
#else
    for (uint32_t G = s >> 6; G <= g1; ++G) {
        const float4 gb = gbox[G];
        if (STATS) st->cand += 1;
        if (box_d2(gb, qe, qx, qy, qz) > best) continue;
        const uint32_t l0 = max(s >> 3, G << 3), l1 = min((e - 1u) >> 3, (G << 3) + 7u);
        for (uint32_t L = l0; L <= l1; ++L) {
            const float4 lb = lbox[L];
            if (STATS) st->cand += 1;
            if (box_d2(lb, qe, qx, qy, qz) > best) continue;
            const uint32_t a = max(s, L << 3), b = min(e, (L << 3) + 8u);
            if (STATS) st->cand += ((b - a + 3) / 4) * 4;
            nn_scan_range<float, PtF, 0, 4>(pts, a, b, qx, qy, qz, best, bj, borig, nullptr);
        }
    }
#endif
}


// ---- the search, in three pieces so that kernels can regroup lanes between them -----------------
// Per-query geometry relative to the grid.
template <typename Real>
struct NNCell {
    int cx, cy, cz;        // the query's cell (may lie outside the grid)
    Real fx, fy, fz;       // offsets of the query inside that cell, in [0, h)
    Real fmin_;            // distance to the nearest face of that cell
    int k0, kmax;          // first ring that can touch the grid box / last ring worth visiting
    Real reach0;           // ring 0 looked at everything within fmin_ + reach0 (the halo margin when the
                           // cell's extended list was scanned, else 0)
};

template <typename Real>
__device__ __forceinline__ NNCell<Real> nn_cell(const Geom<Real> &g, Real qx, Real qy, Real qz, Real bound2) {
    typedef RealTraits<Real> RT;
    NNCell<Real> c;
    const Real lim = (Real)1.0e9;
    Real rx = (qx - g.ox) * g.inv_h, ry = (qy - g.oy) * g.inv_h, rz = (qz - g.oz) * g.inv_h;
    rx = fmin(fmax(rx, -lim), lim); ry = fmin(fmax(ry, -lim), lim); rz = fmin(fmax(rz, -lim), lim);
    c.cx = (int)RT::floor_(rx); c.cy = (int)RT::floor_(ry); c.cz = (int)RT::floor_(rz);
    c.fx = (qx - g.ox) - (Real)c.cx * g.h; c.fy = (qy - g.oy) - (Real)c.cy * g.h; c.fz = (qz - g.oz) - (Real)c.cz * g.h;
    c.fmin_ = fmin(fmin(fmin(c.fx, g.h - c.fx), fmin(c.fy, g.h - c.fy)), fmin(c.fz, g.h - c.fz));
    c.reach0 = (Real)0;
    c.k0 = max(max(max(-c.cx, c.cx - (g.nx - 1)), max(-c.cy, c.cy - (g.ny - 1))), max(max(-c.cz, c.cz - (g.nz - 1)), 0));
    c.kmax = max(max(max(c.cx, g.nx - 1 - c.cx), max(c.cy, g.ny - 1 - c.cy)), max(c.cz, g.nz - 1 - c.cz));
    if (bound2 < RT::inf()) {
        const Real kr = RT::sqrt_rn(bound2) * g.inv_h + (Real)2;
        if (kr < (Real)c.kmax) c.kmax = (int)kr;
    }
    return c;
}

// Ring 0 (the query's own cell) and the empty-space shortcut, from ONE pair of loads (the cell's
// range and its gap field).  Returns the first ring that still has to be visited (>= 1).
// With a halo (point targets): the cell's EXTENDED list is scanned instead -- its own points plus
// every point of the 26 neighbours that lies within `halo` of the shared face / edge / corner.  Any
// point that is not on that list is farther than fmin_ + halo from a query inside the cell, so a
// nearly converged query (residual << halo) is certified by ring 0 alone and never enters the ring
// loop: without the halo the ~8 % of lanes that sit closer to a face than to their match drag their
// whole wave through ring 1 (measured: 75 % of the wave time at the converged pose).
template <typename Real, typename PT, bool STATS = false, bool HALO = false, int TRACK = 0, int B = PCR_NN_BATCH, bool LB = false>
__device__ __forceinline__ int nn_ring0(const Geom<Real> &g, const PT *__restrict__ pts, const uint32_t *__restrict__ cs,
                                        NNCell<Real> &c, Real qx, Real qy, Real qz,
                                        Real &best, uint32_t &bj, uint32_t &borig, NNStats *st = nullptr,
                                        NNTrack<Real> *tk = nullptr) {
    if (c.k0 != 0) return c.k0;                   // outside the grid box: rings below k0 hold no cells
    const uint32_t own = ((uint32_t)c.cz * (uint32_t)g.ny + (uint32_t)c.cy) * (uint32_t)g.nx + (uint32_t)c.cx;
    constexpr bool ext = HALO;                   // (HALO launches only happen for targets that have the lists)
    const uint32_t *__restrict__ csr = ext ? g.cs_h : cs;
    const uint32_t w0 = csr[own], w1 = csr[own + 1];
    const int gap = g.cs_mask != 0xffffffffu ? (int)(w0 >> PCR_GAP_SHIFT) : 0;
    if (ext) {
        // the extended list holds COPIES: track the position in it, then translate the winner to its
        // cell-sorted index (j_h is laid out like the lists, so neighbouring queries share its lines).
        // An EMPTY cell may have a list too (its neighbours' points within the halo): a far-pose query that
        // landed one cell off the surface is served like one inside an occupied cell.
        const uint32_t s_ = w0 & g.cs_mask, e_ = w1 & g.cs_mask;
        if (e_ > s_) {
            if (STATS) { st->rings++; st->rows_loaded++; st->cand += LB ? 0u : ((e_ - s_ + PCR_NN_BATCH - 1) / PCR_NN_BATCH) * PCR_NN_BATCH; }
            uint32_t ej = PCR_NONE;
            if constexpr (LB) nn_scan_range_lb<B, STATS>((const PtF *)g.pts_h, g.lbox_h, g.gbox_h, g.h * 0.00390625f, s_, e_, qx, qy, qz, best, ej, borig, st);
            else nn_scan_range<Real, PT, TRACK, B>((const PT *)g.pts_h, s_, e_, qx, qy, qz, best, ej, borig, tk);
            if (ej != PCR_NONE) bj = g.j_h[ej];
            c.reach0 = g.halo;
            // a whole-cell halo: the list held every point of rings 0 and 1; rings closer than the gap are empty anyway
            const int base = g.halo >= g.h ? 2 : 1;
            return gap > base ? gap : base;
        }
    } else if (gap == 0) {
        const uint32_t s_ = w0 & g.cs_mask, e_ = w1 & g.cs_mask;
        if (STATS) { st->rings++; st->rows_loaded++; st->cand += LB ? 0u : ((e_ - s_ + PCR_NN_BATCH - 1) / PCR_NN_BATCH) * PCR_NN_BATCH; }
        if constexpr (LB) nn_scan_range_lb<B, STATS>(pts, g.lbox, g.gbox, g.h * 0.00390625f, s_, e_, qx, qy, qz, best, bj, borig, st);
        else nn_scan_range<Real, PT, TRACK, B>(pts, s_, e_, qx, qy, qz, best, bj, borig, tk);
        return 1;
    }
    if (g.seed) {                                 // a real point nearby bounds the search from the start
        const uint32_t j0 = g.seed[own];
        if (j0 != PCR_NONE) {
            nn_test<Real, PT, TRACK>(pts[j0], j0, qx, qy, qz, best, bj, borig, tk);
            if (TRACK) nn_track_refresh<Real, TRACK>(*tk, best);
        }
    }
    return gap;                                   // rings closer than `gap` are empty
}

#ifndef PCR_OCC_MIN_RING
#define PCR_OCC_MIN_RING 1
#endif

// true when ring `k` (>= 1) and everything beyond it cannot improve on `best`
template <typename Real>
__device__ __forceinline__ bool nn_certified(const Geom<Real> &g, const NNCell<Real> &c, int k, Real best) {
    if (k > c.kmax) return true;
    const Real lb = (Real)(k - 1) * g.h + c.fmin_ + (k == 1 ? c.reach0 : (Real)0) - g.slack;
    return lb > (Real)0 && lb * lb > best;
}

// Rings kstart (>= 1) .. kmax.
// (a tracking search prunes with tk->prune wherever the plain one prunes with best: PB below)
template <typename Real, typename PT, bool STATS = false, int TRACK = 0, int B = PCR_NN_BATCH, bool LB = false>
__device__ __forceinline__ void nn_rings(const Geom<Real> &g, const PT *__restrict__ pts, const uint32_t *__restrict__ cs,
                                         const NNCell<Real> &c, int kstart, Real qx, Real qy, Real qz,
                                         Real &best, uint32_t &bj, uint32_t &borig, NNStats *st = nullptr,
                                         NNTrack<Real> *tk = nullptr) {
#define PB (TRACK ? tk->prune : best)
    typedef RealTraits<Real> RT;
    const Real lim = (Real)1.0e9;
    const int cx = c.cx, cy = c.cy, cz = c.cz;
    const Real fx = c.fx, fy = c.fy, fz = c.fz;
    const uint32_t unx = (uint32_t)g.nx, plane = (uint32_t)g.ny * (uint32_t)g.nx;   // ncells < 2^32
    for (int k = kstart; k <= c.kmax; ++k) {
        if (nn_certified<Real>(g, c, k, PB)) break;
        if (STATS) st->rings++;
        const int zlo = max(cz - k, 0), zhi = min(cz + k, g.nz - 1);
        const int ylo = max(cy - k, 0), yhi = min(cy + k, g.ny - 1);
        const int xlo = max(cx - k, 0), xhi = min(cx + k, g.nx - 1);
        const int xa = cx - k, xb = cx + k;
        const bool xa_in = xa >= 0 && xa < g.nx, xb_in = xb >= 0 && xb < g.nx;
        Real dxa = fmax((Real)(k - 1) * g.h + fx - g.slack, (Real)0), dxb = fmax((Real)k * g.h - fx - g.slack, (Real)0);
        dxa *= dxa; dxb *= dxb;
        for (int z = zlo; z <= zhi; ++z) {
            const int dzc = z - cz;
            Real dzm = dzc == 0 ? (Real)0 : (dzc > 0 ? (Real)dzc * g.h - fz : (Real)(-dzc - 1) * g.h + fz);
            dzm = fmax(dzm - g.slack, (Real)0);
            const Real dz2 = dzm * dzm;
            if (dz2 > PB) continue;
            const bool zshell = (dzc == k) || (dzc == -k);
            uint32_t row = (uint32_t)z * plane + (uint32_t)ylo * unx;
            for (int y = ylo; y <= yhi; ++y, row += unx) {
                const int dyc = y - cy;
                Real dym = dyc == 0 ? (Real)0 : (dyc > 0 ? (Real)dyc * g.h - fy : (Real)(-dyc - 1) * g.h + fy);
                dym = fmax(dym - g.slack, (Real)0);
                const Real dyz2 = dz2 + dym * dym;
                if (dyz2 > PB) { if (STATS) st->rows_pruned++; continue; }
                if (zshell || dyc == k || dyc == -k) {
                    int xl = xlo, xh = xhi;
                    if (PB < RT::inf()) {                   // clip the row to the remaining budget
                        // approximate sqrt is fine here: the clip only has to be conservative
                        const Real xr = RT::sqrt_fast(PB - dyz2) * (Real)1.000002 + g.slack;
                        const Real a = (qx - xr - g.ox) * g.inv_h, b = (qx + xr - g.ox) * g.inv_h;
                        if (a > (Real)xl) xl = (int)RT::floor_(fmin(a, lim));
                        if (b < (Real)xh) xh = (int)RT::floor_(fmax(b, -lim));
                    }
                    if (xl <= xh) {
                        const uint32_t s_ = cs[row + (uint32_t)xl] & g.cs_mask, e_ = cs[row + (uint32_t)xh + 1u] & g.cs_mask;
                        if (STATS) { st->rows_loaded++; st->cand += LB ? 0u : ((e_ - s_ + PCR_NN_BATCH - 1) / PCR_NN_BATCH) * PCR_NN_BATCH; }
                        if constexpr (LB) nn_scan_range_lb<B, STATS>(pts, g.lbox, g.gbox, g.h * 0.00390625f, s_, e_, qx, qy, qz, best, bj, borig, st); else nn_scan_range<Real, PT, TRACK, B>(pts, s_, e_, qx, qy, qz, best, bj, borig, tk);
                    }
                } else {                                    // interior row of the ring: its two end cells
                    if (xa_in && dyz2 + dxa <= PB) {
                        const uint32_t s_ = cs[row + (uint32_t)xa] & g.cs_mask, e_ = cs[row + (uint32_t)xa + 1u] & g.cs_mask;
                        if (STATS) { st->rows_loaded++; st->cand += LB ? 0u : ((e_ - s_ + PCR_NN_BATCH - 1) / PCR_NN_BATCH) * PCR_NN_BATCH; }
                        if constexpr (LB) nn_scan_range_lb<B, STATS>(pts, g.lbox, g.gbox, g.h * 0.00390625f, s_, e_, qx, qy, qz, best, bj, borig, st); else nn_scan_range<Real, PT, TRACK, B>(pts, s_, e_, qx, qy, qz, best, bj, borig, tk);
                    }
                    if (xb_in && dyz2 + dxb <= PB) {
                        const uint32_t s_ = cs[row + (uint32_t)xb] & g.cs_mask, e_ = cs[row + (uint32_t)xb + 1u] & g.cs_mask;
                        if (STATS) { st->rows_loaded++; st->cand += LB ? 0u : ((e_ - s_ + PCR_NN_BATCH - 1) / PCR_NN_BATCH) * PCR_NN_BATCH; }
                        if constexpr (LB) nn_scan_range_lb<B, STATS>(pts, g.lbox, g.gbox, g.h * 0.00390625f, s_, e_, qx, qy, qz, best, bj, borig, st); else nn_scan_range<Real, PT, TRACK, B>(pts, s_, e_, qx, qy, qz, best, bj, borig, tk);
                    }
                }
            }
        }
    }
#undef PB
}

// ---- rings with TIGHT row-block boxes (round 6; float32 point targets that carry Geom::rbox) ----------------------------
// The plain ring loop bounds a row of cells from below by the distance to the cells' CUBES.  A cloud is a set of surfaces: the
// ground fills a few centimetres of the 40-cm layer of cells it lies in, so a query D above it sees "D - (up to) h" for
// every cell of that layer, visits every row within sqrt(2 D h) of its foot point and tests their points (82 candidates and
// 7.2 row segments per query at the first pose of plane_b01, of which the ball really contains a handful).  Here every
// segment of a row first reads the box of the POINTS in its 8-cell block (one 8-byte record out of a table of 1 byte per cell:
// cache-resident where cell_start is not), is skipped when that box lies beyond the current best, and otherwise is clipped
// in x to the block's occupied cells and to what the ball leaves of the box's (y, z) distance.  Bounds stay conservative: a
// bound byte stands for [b, b + 1), and `mq` (grid slack + 1.5 quantisation steps) is taken off every side.
template <bool STATS = false, int B = PCR_NN_BATCH>
__device__ __forceinline__ void nn_rings_box(const Geom<float> &g, const PtF *__restrict__ pts, const uint32_t *__restrict__ cs,
                                             const NNCell<float> &c, int kstart, float qx, float qy, float qz,
                                             float &best, uint32_t &bj, uint32_t &borig, NNStats *st = nullptr) {
    typedef RealTraits<float> RT;
    const float lim = 1.0e9f;
    const int cx = c.cx, cy = c.cy, cz = c.cz;
    const float fx = c.fx, fy = c.fy, fz = c.fz;
    const uint32_t unx = (uint32_t)g.nx, plane = (uint32_t)g.ny * (uint32_t)g.nx;   // ncells < 2^32
    const float qs = 256.f * g.inv_h;                   // quantisation steps per metre (y, z; an x step is 8 of them)
    const float qs2 = qs * qs;
    const float mq = g.slack * qs + 1.5f;
    const float ux0 = (qx - g.ox) * qs, uy0 = (qy - g.oy) * qs, uz0 = (qz - g.oz) * qs;
    // Variant B (the one kept in the tree; variant A -- commit 55c01ec -- read the box FIRST and clipped the segment to the
    // block's occupied cells and to the ball's reach: a third dependent load per row, slower at every pose): the box records
    // of a shell row's segment [xl, xh] are requested TOGETHER with its two cell_start words, and the segment's candidates are
    // skipped when every non-empty block it touches lies beyond the current best.  Returns true when the segment must be read.
    auto box_hit = [&](uint32_t rrow, float uy, float uz, int xl, int xh) __attribute__((always_inline)) -> bool {
        bool hit = false;
        const float pbq = best * qs2;
        for (int xb = xl >> PCR_RB_LOG; xb <= (xh >> PCR_RB_LOG); ++xb) {
            const uint2 w = g.rbox[rrow + (uint32_t)xb];
            const int x0 = xb << PCR_RB_LOG;
            const int lo = max(xl - x0, 0), hi = min(xh - x0, (1 << PCR_RB_LOG) - 1);
            const uint32_t m = (w.x & 0xffu) & (0xffu << lo) & (0xffu >> (7 - hi));
            const float ux = ux0 - (float)x0 * 256.f;                            // the query from the block's corner, y / z steps
            const float xlo = (float)((w.x >> 8) & 0xffu) * 8.f, xhi = (float)((w.x >> 16) & 0xffu) * 8.f + 8.f;
            const float ylo = (float)(w.x >> 24), yhi = (float)(w.y & 0xffu) + 1.f;
            const float zlo = (float)((w.y >> 8) & 0xffu), zhi = (float)((w.y >> 16) & 0xffu) + 1.f;
            const float dx = fmaxf(fmaxf(xlo - ux, ux - xhi) - mq, 0.f);
            const float dy = fmaxf(fmaxf(ylo - uy, uy - yhi) - mq, 0.f);
            const float dz = fmaxf(fmaxf(zlo - uz, uz - zhi) - mq, 0.f);
            hit |= (m != 0) & !((dy * dy + dz * dz) + dx * dx > pbq);
        }
        return hit;
    };
    for (int k = kstart; k <= c.kmax; ++k) {
        if (nn_certified<float>(g, c, k, best)) break;
        if (STATS) st->rings++;
        const int zlo = max(cz - k, 0), zhi = min(cz + k, g.nz - 1);
        const int ylo = max(cy - k, 0), yhi = min(cy + k, g.ny - 1);
        const int xlo = max(cx - k, 0), xhi = min(cx + k, g.nx - 1);
        const int xa = cx - k, xb = cx + k;
        const bool xa_in = xa >= 0 && xa < g.nx, xb_in = xb >= 0 && xb < g.nx;
        float dxa = fmaxf((float)(k - 1) * g.h + fx - g.slack, 0.f), dxb = fmaxf((float)k * g.h - fx - g.slack, 0.f);
        dxa *= dxa; dxb *= dxb;
        for (int z = zlo; z <= zhi; ++z) {
            const int dzc = z - cz;
            float dzm = dzc == 0 ? 0.f : (dzc > 0 ? (float)dzc * g.h - fz : (float)(-dzc - 1) * g.h + fz);
            dzm = fmaxf(dzm - g.slack, 0.f);
            const float dz2 = dzm * dzm;
            if (dz2 > best) continue;
            const bool zshell = (dzc == k) || (dzc == -k);
            const float uz = uz0 - (float)z * 256.f;
            uint32_t row = (uint32_t)z * plane + (uint32_t)ylo * unx;
            uint32_t rrow = ((uint32_t)z * (uint32_t)g.ny + (uint32_t)ylo) * (uint32_t)g.nxr;
            for (int y = ylo; y <= yhi; ++y, row += unx, rrow += (uint32_t)g.nxr) {
                const int dyc = y - cy;
                float dym = dyc == 0 ? 0.f : (dyc > 0 ? (float)dyc * g.h - fy : (float)(-dyc - 1) * g.h + fy);
                dym = fmaxf(dym - g.slack, 0.f);
                const float dyz2 = dz2 + dym * dym;
                if (dyz2 > best) { if (STATS) st->rows_pruned++; continue; }
                if (zshell || dyc == k || dyc == -k) {
                    int xl = xlo, xh = xhi;
                    if (best < RT::inf()) {                 // clip the row to the remaining budget (cube bound; the box clips again)
                        const float xr = RT::sqrt_fast(best - dyz2) * 1.000002f + g.slack;
                        const float a = (qx - xr - g.ox) * g.inv_h, b = (qx + xr - g.ox) * g.inv_h;
                        if (a > (float)xl) xl = (int)RT::floor_(fminf(a, lim));
                        if (b < (float)xh) xh = (int)RT::floor_(fmaxf(b, -lim));
                    }
                    if (xl <= xh) {
                        const uint32_t s_ = cs[row + (uint32_t)xl] & g.cs_mask, e_ = cs[row + (uint32_t)xh + 1u] & g.cs_mask;
                        const bool hit = box_hit(rrow, uy0 - (float)y * 256.f, uz, xl, xh);
                        if (hit) {
                            if (STATS) { st->rows_loaded++; st->cand += ((e_ - s_ + PCR_NN_BATCH - 1) / PCR_NN_BATCH) * PCR_NN_BATCH; }
                            nn_scan_range<float, PtF, 0, B>(pts, s_, e_, qx, qy, qz, best, bj, borig, nullptr);
                        } else if (STATS) st->rows_pruned++;
                    }
                } else {                                    // interior row of the ring: its two end cells (single cells: no box test)
                    if (xa_in && dyz2 + dxa <= best) {
                        const uint32_t s_ = cs[row + (uint32_t)xa] & g.cs_mask, e_ = cs[row + (uint32_t)xa + 1u] & g.cs_mask;
                        if (STATS) { st->rows_loaded++; st->cand += ((e_ - s_ + PCR_NN_BATCH - 1) / PCR_NN_BATCH) * PCR_NN_BATCH; }
                        nn_scan_range<float, PtF, 0, B>(pts, s_, e_, qx, qy, qz, best, bj, borig, nullptr);
                    }
                    if (xb_in && dyz2 + dxb <= best) {
                        const uint32_t s_ = cs[row + (uint32_t)xb] & g.cs_mask, e_ = cs[row + (uint32_t)xb + 1u] & g.cs_mask;
                        if (STATS) { st->rows_loaded++; st->cand += ((e_ - s_ + PCR_NN_BATCH - 1) / PCR_NN_BATCH) * PCR_NN_BATCH; }
                        nn_scan_range<float, PtF, 0, B>(pts, s_, e_, qx, qy, qz, best, bj, borig, nullptr);
                    }
                }
            }
        }
    }
}

// The same rings with the rows of a slab taken from the row-occupancy bitmap (Geom::rowocc): the centroid search when
// the gate spans >= 5 rings (kernels.hip: k_nn_scan<VOXEL = 2>).  A separate function on purpose: routing the plain
// search through the shared row body (a lambda) cost the 1e8-point search 5-13 % (74.5 / 80.5 vs 70.9 ms per 26 passes).
template <typename Real, typename PT, bool STATS = false, int TRACK = 0>
__device__ __forceinline__ void nn_rings_occ(const Geom<Real> &g, const PT *__restrict__ pts, const uint32_t *__restrict__ cs,
                                         const NNCell<Real> &c, int kstart, Real qx, Real qy, Real qz,
                                         Real &best, uint32_t &bj, uint32_t &borig, NNStats *st = nullptr,
                                         NNTrack<Real> *tk = nullptr) {
#define PB (TRACK ? tk->prune : best)
    typedef RealTraits<Real> RT;
    const Real lim = (Real)1.0e9;
    const int cx = c.cx, cy = c.cy, cz = c.cz;
    const Real fx = c.fx, fy = c.fy, fz = c.fz;
    const uint32_t unx = (uint32_t)g.nx, plane = (uint32_t)g.ny * (uint32_t)g.nx;   // ncells < 2^32
    for (int k = kstart; k <= c.kmax; ++k) {
        if (nn_certified<Real>(g, c, k, PB)) break;
        if (STATS) st->rings++;
        const int zlo = max(cz - k, 0), zhi = min(cz + k, g.nz - 1);
        const int ylo = max(cy - k, 0), yhi = min(cy + k, g.ny - 1);
        const int xlo = max(cx - k, 0), xhi = min(cx + k, g.nx - 1);
        const int xa = cx - k, xb = cx + k;
        const bool xa_in = xa >= 0 && xa < g.nx, xb_in = xb >= 0 && xb < g.nx;
        Real dxa = fmax((Real)(k - 1) * g.h + fx - g.slack, (Real)0), dxb = fmax((Real)k * g.h - fx - g.slack, (Real)0);
        dxa *= dxa; dxb *= dxb;
        for (int z = zlo; z <= zhi; ++z) {
            const int dzc = z - cz;
            Real dzm = dzc == 0 ? (Real)0 : (dzc > 0 ? (Real)dzc * g.h - fz : (Real)(-dzc - 1) * g.h + fz);
            dzm = fmax(dzm - g.slack, (Real)0);
            const Real dz2 = dzm * dzm;
            if (dz2 > PB) continue;
            const bool zshell = (dzc == k) || (dzc == -k);
            // one row of cells (y, z): the row's distance, then its cells inside the ring
            auto do_row = [&](int y) __attribute__((always_inline)) {
                const uint32_t row = (uint32_t)z * plane + (uint32_t)y * unx;
                const int dyc = y - cy;
                Real dym = dyc == 0 ? (Real)0 : (dyc > 0 ? (Real)dyc * g.h - fy : (Real)(-dyc - 1) * g.h + fy);
                dym = fmax(dym - g.slack, (Real)0);
                const Real dyz2 = dz2 + dym * dym;
                if (dyz2 > PB) { if (STATS) st->rows_pruned++; return; }
                if (zshell || dyc == k || dyc == -k) {
                    int xl = xlo, xh = xhi;
                    if (PB < RT::inf()) {                   // clip the row to the remaining budget
                        // approximate sqrt is fine here: the clip only has to be conservative
                        const Real xr = RT::sqrt_fast(PB - dyz2) * (Real)1.000002 + g.slack;
                        const Real a = (qx - xr - g.ox) * g.inv_h, b = (qx + xr - g.ox) * g.inv_h;
                        if (a > (Real)xl) xl = (int)RT::floor_(fmin(a, lim));
                        if (b < (Real)xh) xh = (int)RT::floor_(fmax(b, -lim));
                    }
                    if (xl <= xh) {
                        const uint32_t s_ = cs[row + (uint32_t)xl] & g.cs_mask, e_ = cs[row + (uint32_t)xh + 1u] & g.cs_mask;
                        if (STATS) { st->rows_loaded++; st->cand += ((e_ - s_ + PCR_NN_BATCH - 1) / PCR_NN_BATCH) * PCR_NN_BATCH; }
                        nn_scan_range<Real, PT, TRACK>(pts, s_, e_, qx, qy, qz, best, bj, borig, tk);
                    }
                } else {                                    // interior row of the ring: its two end cells
                    if (xa_in && dyz2 + dxa <= PB) {
                        const uint32_t s_ = cs[row + (uint32_t)xa] & g.cs_mask, e_ = cs[row + (uint32_t)xa + 1u] & g.cs_mask;
                        if (STATS) { st->rows_loaded++; st->cand += ((e_ - s_ + PCR_NN_BATCH - 1) / PCR_NN_BATCH) * PCR_NN_BATCH; }
                        nn_scan_range<Real, PT, TRACK>(pts, s_, e_, qx, qy, qz, best, bj, borig, tk);
                    }
                    if (xb_in && dyz2 + dxb <= PB) {
                        const uint32_t s_ = cs[row + (uint32_t)xb] & g.cs_mask, e_ = cs[row + (uint32_t)xb + 1u] & g.cs_mask;
                        if (STATS) { st->rows_loaded++; st->cand += ((e_ - s_ + PCR_NN_BATCH - 1) / PCR_NN_BATCH) * PCR_NN_BATCH; }
                        nn_scan_range<Real, PT, TRACK>(pts, s_, e_, qx, qy, qz, best, bj, borig, tk);
                    }
                }
            };
            // Which rows?  Every y of the ring -- or (OCC: the centroid search when the gate spans >= 5 rings), for the rings of a far query
            // (k >= PCR_OCC_MIN_RING) only the rows with a point in the 16-cell x-blocks around the query: one 8-byte load
            // answers it for 64 rows, where the row loop spends two cell_start loads on every empty row.  Measured per
            // pass: centroid search, vplane_10m (0.5 m voxels) 1.19 -> 0.96 ms at the converged poses, ndt_10m (1 m voxels)
            // 0.62 -> 0.64; the float32 point search LOSES
            // (89-91 instead of 80 VGPRs, 5 waves/SIMD, and the bit iteration: plane_b01 +8 %, 1e8 points +10 %), so it
            // keeps the plain row loop.
            {
                const bool occ = g.rowocc != nullptr && k >= PCR_OCC_MIN_RING;
                const int xb0 = xlo >> 4, xb1 = xhi >> 4;
                for (int wy = ylo >> 6; wy <= (yhi >> 6); ++wy) {
                    const int lo = max(ylo - 64 * wy, 0), hi = min(yhi - 64 * wy, 63);
                    unsigned long long m = (~0ull << lo) & (~0ull >> (63 - hi));
                    if (occ) {
                        const unsigned long long *wp = g.rowocc + ((size_t)z * (size_t)g.nyw + (size_t)wy) * (size_t)g.nxb;
                        unsigned long long o = 0;
                        for (int xb_ = xb0; xb_ <= xb1; ++xb_) o |= wp[xb_];
                        m &= o;
                    }
                    while (m) {
                        const int b = __builtin_ctzll(m);
                        m &= m - 1;
                        do_row(64 * wy + b);
                    }
                }
            }
        }
    }
#undef PB
}

// On return: borig = ORIGINAL index of the nearest point (PCR_NONE if nothing closer than
// sqrt(bound2)), best = its squared distance, bj = its cell-sorted index.
// SEEDED: best / bj / borig come in holding a real target point (any point is an exact upper bound:
// the search then only has to look inside that radius) or (bound2, PCR_NONE, PCR_NONE).
// TRACK: `tk` was initialised with nn_track_init(tk, bound2, mu); on return min(tk->second, tk->pmin) is a lower
// bound on the squared distance to every target point other than the winner (to every point if there is none).
template <typename Real, typename PT, bool STATS = false, bool SEEDED = false, bool HALO = false, int TRACK = 0, bool OCC = false,
          int B = PCR_NN_BATCH, bool RBOX = false, bool LB = false>
__device__ __forceinline__ void nn_search(const Geom<Real> &g, const PT *__restrict__ pts,
                                          const uint32_t *__restrict__ cs,
                                          Real qx, Real qy, Real qz, Real bound2,
                                          Real &best, uint32_t &bj, uint32_t &borig, NNStats *st = nullptr,
                                          NNTrack<Real> *tk = nullptr) {
    if (!SEEDED) { best = bound2; bj = PCR_NONE; borig = PCR_NONE; }
    NNCell<Real> c = nn_cell<Real>(g, qx, qy, qz, bound2);
    static_assert(!LB || (sizeof(Real) == 4 && TRACK == 0 && !OCC && !RBOX), "leaf / group boxes: plain float32 point search only");
    const int kstart = nn_ring0<Real, PT, STATS, HALO, TRACK, B, LB>(g, pts, cs, c, qx, qy, qz, best, bj, borig, st, tk);
    if constexpr (LB) {
        nn_rings<Real, PT, STATS, TRACK, B, true>(g, pts, cs, c, kstart, qx, qy, qz, best, bj, borig, st, tk);
    } else if constexpr (RBOX) {
        static_assert(sizeof(Real) == 4 && TRACK == 0 && !OCC, "row-block boxes: plain float32 point search only");
        nn_rings_box<STATS, B>(g, pts, cs, c, kstart, qx, qy, qz, best, bj, borig, st);
    } else if (OCC) nn_rings_occ<Real, PT, STATS, TRACK>(g, pts, cs, c, kstart, qx, qy, qz, best, bj, borig, st, tk);
    else nn_rings<Real, PT, STATS, TRACK, B>(g, pts, cs, c, kstart, qx, qy, qz, best, bj, borig, st, tk);
}

