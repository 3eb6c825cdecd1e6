// Wave-cooperative exact nearest-neighbour search with an MFMA distance FILTER (round 5; far poses of a point target).
//
// Why.  The per-lane ring search (nn_device.h) is at a plateau no one-sided change moves (DESIGN / docs/EXPERIMENTS.md): at the
// first poses of a Gauss-Newton run a query tests ~84 candidates and touches ~47 cache lines, and neither a perfect initial
// bound (round 5: 212 vs 218 us with the true neighbour as seed) nor fewer load instructions, more waves, another cell
// size or LDS staging changes that.  What does is testing candidates at a fraction of the VALU cost and without any
// per-lane address stream: the 64 queries of a tile are Morton neighbours moved by one rigid transform, the union of their
// search balls is one small box of cells, and  |q - c|^2 = |q|^2 - 2 q.c + |c|^2  over (64 queries) x (32 candidates) is a
// K = 4 contraction -- two v_mfma_f32_32x32x2_f32 per 32 x 32 block, 8 pairs per cycle and SIMD where nn_test's 12 VALU
// instructions manage 1.3.
//
// Exactness.  The MFMA distances are a FILTER only (coordinates relative to the box centre, |error| <= err, bounded below):
//   1. every lane takes an exact upper bound from a real point near its cell (the cell's first point / the seed of an
//      empty cell); the wave's box = union of the balls through those points, so it contains every lane's true neighbour;
//   2. all candidates of the box's rows go through the MFMA; each lane keeps, for the two queries it serves (see the
//      layout below) and its half of the candidate rows, the smallest and the second smallest GROUP minimum (a group = 4
//      consecutive records) and where the smallest came from;
//   3. the 4 records of the winning group are tested with the exact arithmetic of the per-lane search (nn_test: float32
//      fma distance, ties to the smaller original index); every other candidate of the box is no closer than the second
//      group minimum minus the error bound.  If that is strictly above the exact winner's distance the winner is what the
//      per-lane search returns -- otherwise (two candidates within ~2e-6 m^2 of each other, duplicated points, a ball too
//      large for a shared box) the lane runs the per-lane search, seeded with what it has.
// The results are therefore bit-identical to k_nn_scan's by construction; the parity tests run both.
//
// Layout of v_mfma_f32_32x32x2_f32 (A 32 x 2, B 2 x 32, D 32 x 32; guide section 3): lane l supplies A[i = l & 31][k = l >> 5]
// and B[k = l >> 5][j = l & 31] and receives D[row = (r & 3) + 8 (r >> 2) + 4 (l >> 5)][col = l & 31] in register r.
// Rows = candidates, columns = queries: a lane's 16 results all belong to ONE query (no cross-lane work to find a minimum),
// registers 4k .. 4k+3 are 4 CONSECUTIVE candidates (one group, one 64-byte line).  Two column blocks cover the 64 queries:
// block 0 = the queries of lanes 0..31, block 1 = lanes 32..63; lane l works for query (l & 31) of either block on the rows
// of its half, and the two halves merge with one exchange at the end.
#pragma once

#include "pass_device.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
extern "C" __device__ int __ockl_wfred_min_i32(int);
extern "C" __device__ int __ockl_wfred_max_i32(int);
extern "C" __device__ unsigned __ockl_wfred_add_u32(unsigned);
extern "C" __device__ float __ockl_wfred_max_f32(float);

#ifndef PCR_MF_MAXC
#define PCR_MF_MAXC 2048         // candidates of a box (64 MFMA tiles); beyond that the tile's points go to the per-lane search
#endif
#ifndef PCR_MF_RCAP
#define PCR_MF_RCAP 3.6f         // a lane whose initial ball is wider than this many cells does not join the shared box
#endif
#ifndef PCR_MF_PIPE
#define PCR_MF_PIPE 4            // candidate tiles whose records are in flight during the sweep
#endif
#define PCR_MF_NONE 0xffffffffu

#ifdef PCR_MF_STATS
// developer counters (tools/build_variant.sh mfstats "-DPCR_MF_STATS=1"; read with pcr_mf_stats_read): [0] query tiles, [1] tiles
// that shared a box, [2] candidate tiles swept, [3] rows of the boxes, [4] rows swept, [5] lanes outside the box (ball too wide / no
// bound), [6] lanes the filter could not certify, [7] flushes of a full deferred list
__device__ unsigned long long g_mf_stats[16];      // [8..14]: wave cycles in the phases (seed, box, list, sweep, exact + merge, flush, whole tile)
#define MF_STAT(k, v) do { if ((threadIdx.x & 63) == 0) atomicAdd(&g_mf_stats[k], (unsigned long long)(v)); } while (0)
#define MF_CLK(var) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long var = __builtin_readcyclecounter()
#else
#define MF_STAT(k, v) do { } while (0)
#define MF_CLK(var) do { } while (0)
#endif

struct MfBox { int x0, x1, y0, y1, z0, z1; };

// cells a ball of radius r around the query can reach, clamped to the grid (r carries the rounding slack)
__device__ __forceinline__ MfBox mf_ball_cells(const Geom<float> &g, float tx, float ty, float tz, float r) {
    MfBox b;
    const float fx = (float)(g.nx - 1), fy = (float)(g.ny - 1), fz = (float)(g.nz - 1);
    b.x0 = (int)fminf(fmaxf(floorf((tx - r - g.ox) * g.inv_h), 0.f), fx);
    b.x1 = (int)fminf(fmaxf(floorf((tx + r - g.ox) * g.inv_h), 0.f), fx);
    b.y0 = (int)fminf(fmaxf(floorf((ty - r - g.oy) * g.inv_h), 0.f), fy);
    b.y1 = (int)fminf(fmaxf(floorf((ty + r - g.oy) * g.inv_h), 0.f), fy);
    b.z0 = (int)fminf(fmaxf(floorf((tz - r - g.oz) * g.inv_h), 0.f), fz);
    b.z1 = (int)fminf(fmaxf(floorf((tz + r - g.oz) * g.inv_h), 0.f), fz);
    return b;
}

// what a lane tracks for one query block over its half of the candidate rows: packed group minima -- the bits of the
// (positive) approximate squared distance with the group number in the two low bits -- order like the values
struct MfTrack {
    uint32_t b1, b2;       // smallest and second smallest group minimum seen so far
    uint32_t t1;           // candidate tile b1 came from
};

__device__ __forceinline__ uint32_t mf_umin(uint32_t a, uint32_t b) { return a < b ? a : b; }
__device__ __forceinline__ uint32_t mf_umax(uint32_t a, uint32_t b) { return a > b ? a : b; }

// acc = |c'|^2 - 2 q'.c' for 16 candidates of one query; n0 = |q'|^2 + bias makes it the (positive) approximate distance
__device__ __forceinline__ void mf_track(MfTrack &m, const f32x16 &acc, float n0, uint32_t t) {
    uint32_t gk[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float gm = fminf(fminf(acc[4 * k], acc[4 * k + 1]), fminf(acc[4 * k + 2], acc[4 * k + 3])) + n0;
        gk[k] = (__float_as_uint(gm) & ~3u) | (uint32_t)k;
    }
    // the two smallest of the four
    const uint32_t lo1 = mf_umin(gk[0], gk[1]), hi1 = mf_umax(gk[0], gk[1]);
    const uint32_t lo2 = mf_umin(gk[2], gk[3]), hi2 = mf_umax(gk[2], gk[3]);
    const uint32_t m1 = mf_umin(lo1, lo2), m2 = mf_umin(mf_umax(lo1, lo2), mf_umin(hi1, hi2));
    // second smallest of {b1 <= b2, m1 <= m2} = min(max(b1, m1), min(b2, m2))
    const bool better = m1 < m.b1;
    m.b2 = mf_umin(mf_umax(m.b1, m1), mf_umin(m.b2, m2));
    m.t1 = better ? t : m.t1;
    m.b1 = mf_umin(m.b1, m1);
}

struct MfRes {             // exact result of one block on this lane's half of the rows
    float d;               // squared distance of the best tested record (bound2 if none)
    uint32_t j, o;         // its cell-sorted / original index
    float s;               // lower bound (approximate, before the error margin) on every record of this half NOT tested exactly
};

__device__ __forceinline__ float mf_shfl32(float v) { return __shfl_xor(v, 32, 64); }
__device__ __forceinline__ uint32_t mf_shfl32(uint32_t v) { return (uint32_t)__shfl_xor((int)v, 32, 64); }

// Points the filter could not settle (a ball too wide for a shared box, two candidates the error bound cannot separate, a box
// with too many rows or records) are NOT searched where they turn up -- one such lane would drag its whole wave through a
// per-lane search, and at the first pose every second tile has one -- but collected per wave in LDS and searched 64 at a
// time by nn_mfma_flush: dense waves, a handful of searches per wave instead of one per tile.
struct MfWave {
    uint32_t *cidx;        // [PCR_MF_MAXC] cell-sorted indices of the box's candidates, rows packed back to back
    uint32_t *defer;       // [64] scan points left to the per-lane search
    int ndefer;            // wave-uniform
};

template <int HALO>
__device__ __forceinline__ void nn_mfma_flush(const LinArgs &a, const Geom<float> &g, const PoseK &P, MfWave &w) {
    const int lane = threadIdx.x & 63;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    MF_CLK(cf0);
    if (lane < w.ndefer) {
        const int64_t i = (int64_t)w.defer[lane];
        float tx, ty, tz;
        xform(P, a.sx[i], a.sy[i], a.sz[i], tx, ty, tz);
        float best; uint32_t bj, bo;
        nn_search<float, PtF, false, false, HALO != 0, 0>(g, a.pts, a.cell_start, tx, ty, tz, a.bound2_f, best, bj, bo);
        const bool okm = bo != PCR_NONE && __builtin_sqrtf(best) < a.md_f;
        a.nn_j[i] = okm ? bj : PCR_NONE;
    }
    __builtin_amdgcn_wave_barrier();
    MF_CLK(cf1);
#ifdef PCR_MF_STATS
    MF_STAT(13, cf1 - cf0);
#endif
    w.ndefer = 0;
}

// One tile of 64 scan points.  n_pad = index of the last readable record of a.pts (the sentinels included).  Writes a.nn_j like
// nn_point<0, HALO, 0>, except for the lanes it appends to w.defer (their nn_j is written by nn_mfma_flush).
// The bound of step 1 as a kernel of its own (PRE): one lane per scan point, nothing but the short dependent chain cell ->
// seed -> point at full occupancy; it leaves the SQUARED distance to a real point near the query's cell in a.lb2 (-1: a
// point that matches nothing -- NaN / inf).  Inside nn_tile_mfma the same chain is three dependent round trips per TILE at
// 3-4 waves per SIMD: a third of the tile's latency.
__device__ __forceinline__ void nn_mfma_bound(const LinArgs &a, const Geom<float> &g, const PoseK &P, int64_t i) {
    float tx, ty, tz;
    xform(P, a.sx[i], a.sy[i], a.sz[i], tx, ty, tz);
    const bool live = fabsf(tx) <= 3.0e38f && fabsf(ty) <= 3.0e38f && fabsf(tz) <= 3.0e38f;
    float best = a.bound2_f;
    uint32_t bj = PCR_NONE, bo = PCR_NONE;
    const NNCell<float> c = nn_cell<float>(g, tx, ty, tz, a.bound2_f);
    if (live && c.k0 == 0) {
        const uint32_t own = ((uint32_t)c.cz * (uint32_t)g.ny + (uint32_t)c.cy) * (uint32_t)g.nx + (uint32_t)c.cx;
        const uint32_t s_ = a.cell_start[own] & g.cs_mask, e_ = a.cell_start[own + 1] & g.cs_mask;
        if (e_ > s_) {
            nn_scan_range<float, PtF, 0>(a.pts, s_, e_, tx, ty, tz, best, bj, bo);       // the cell's own points: a tight ball at the near poses
        } else if (g.seed) {
            const uint32_t js = g.seed[own];
            if (js != PCR_NONE) nn_test<float, PtF, 0>(a.pts[js], js, tx, ty, tz, best, bj, bo);
        }
    }
    a.lb2[i] = live ? best : -1.f;
}

template <int HALO, int PRE>
__device__ __forceinline__ void nn_tile_mfma(const LinArgs &a, const Geom<float> &g, const PoseK &P, MfWave &w,
                                             uint32_t n_pad, int64_t first, int64_t end) {
    typedef RealTraits<float> RT;
    const int lane = threadIdx.x & 63;
    const bool half = lane >= 32;
    const int64_t i = first + lane;
    const bool exists = i < end;
    float x = 0.f, y = 0.f, z = 0.f;
    if (exists) { x = a.sx[i]; y = a.sy[i]; z = a.sz[i]; }
    float tx, ty, tz;
    xform(P, x, y, z, tx, ty, tz);
    // NaN / inf queries match nothing
    const float bq = (PRE && exists) ? a.lb2[i] : 0.f;
    const bool live = exists && fabsf(tx) <= 3.0e38f && fabsf(ty) <= 3.0e38f && fabsf(tz) <= 3.0e38f && bq >= 0.f;
    float best = a.bound2_f;
    uint32_t bj = PCR_NONE, bo = PCR_NONE;
    const NNCell<float> c = nn_cell<float>(g, tx, ty, tz, a.bound2_f);
    MF_CLK(ck0);
    // ---- 1. an exact upper bound from a real point: the first point of the query's cell, or the seed of an empty cell
    if (!PRE && live && c.k0 == 0) {
        const uint32_t own = ((uint32_t)c.cz * (uint32_t)g.ny + (uint32_t)c.cy) * (uint32_t)g.nx + (uint32_t)c.cx;
        const uint32_t s_ = a.cell_start[own] & g.cs_mask, e_ = a.cell_start[own + 1] & g.cs_mask;
        uint32_t js = e_ > s_ ? s_ : (g.seed ? g.seed[own] : PCR_NONE);
        if (js != PCR_NONE) nn_test<float, PtF, 0>(a.pts[js], js, tx, ty, tz, best, bj, bo);
    }
    const float r2 = PRE ? bq : best;                                  // (PRE: the bound came from k_nn_bound; no real point is held)
    const float r = RT::sqrt_fast(r2) * 1.000002f + g.slack;          // (no seed: the ball of the search bound)
    bool coop = live && r <= PCR_MF_RCAP * g.h;
    bool pending = live && !coop;
    MF_STAT(0, 1); MF_STAT(5, __popcll(__ballot(pending)));
    MF_CLK(ck1);
#ifdef PCR_MF_STATS
    MF_STAT(8, ck1 - ck0);
#endif
    if (__any(coop)) {
        // ---- 2. the shared box
        const MfBox b = mf_ball_cells(g, tx, ty, tz, r);
        const int X0 = __ockl_wfred_min_i32(coop ? b.x0 : 0x7fffffff), X1 = __ockl_wfred_max_i32(coop ? b.x1 : -1);
        const int Y0 = __ockl_wfred_min_i32(coop ? b.y0 : 0x7fffffff), Y1 = __ockl_wfred_max_i32(coop ? b.y1 : -1);
        const int Z0 = __ockl_wfred_min_i32(coop ? b.z0 : 0x7fffffff), Z1 = __ockl_wfred_max_i32(coop ? b.z1 : -1);
        const int by = Y1 - Y0 + 1, bz = Z1 - Z0 + 1, rows = by * bz;
        bool ok = by > 0 && bz > 0 && rows <= 128;
        // lane l fetches the point ranges of rows l and l + 64
        uint32_t rs[2] = {0, 0}, re[2] = {0, 0};
        if (ok) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int rw = lane + 64 * q;
                if (rw < rows) {
                    const uint32_t ry = (uint32_t)(Y0 + rw % by), rz = (uint32_t)(Z0 + rw / by);
                    const uint32_t rowb = (rz * (uint32_t)g.ny + ry) * (uint32_t)g.nx;
                    rs[q] = a.cell_start[rowb + (uint32_t)X0] & g.cs_mask;
                    re[q] = a.cell_start[rowb + (uint32_t)X1 + 1u] & g.cs_mask;
                }
            }
        }
        MF_CLK(ck2);
        // ---- 3a. the candidates: rows no ball reaches are dropped, the others listed back to back in LDS
        uint32_t N = 0;
        if (ok) {
            int rr = 0;
            for (int zz = Z0; zz <= Z1; ++zz) {
                const int dzc = zz - c.cz;
                float dzm = dzc == 0 ? 0.f : (dzc > 0 ? (float)dzc * g.h - c.fz : (float)(-dzc - 1) * g.h + c.fz);
                dzm = fmaxf(dzm - g.slack, 0.f);
                const float dz2 = dzm * dzm;
                const bool zneed = __any(coop && dz2 <= r2);
                for (int yy = Y0; yy <= Y1; ++yy, ++rr) {
                    const uint32_t s_ = (uint32_t)__builtin_amdgcn_readlane((int)(rr < 64 ? rs[0] : rs[1]), rr & 63);
                    const uint32_t e_ = (uint32_t)__builtin_amdgcn_readlane((int)(rr < 64 ? re[0] : re[1]), rr & 63);
                    if (s_ == e_ || !zneed) continue;
                    const int dyc = yy - c.cy;
                    float dym = dyc == 0 ? 0.f : (dyc > 0 ? (float)dyc * g.h - c.fy : (float)(-dyc - 1) * g.h + c.fy);
                    dym = fmaxf(dym - g.slack, 0.f);
                    if (!__any(coop && dz2 + dym * dym <= r2)) continue;        // no ball reaches this row
                    MF_STAT(4, 1);
                    const uint32_t len = e_ - s_;
                    if (N + len > PCR_MF_MAXC) { ok = false; break; }
                    for (uint32_t o = (uint32_t)lane; o < len; o += 64u) w.cidx[N + o] = s_ + o;
                    N += len;
                }
                if (!ok) break;
            }
        }
        MF_CLK(ck3);
#ifdef PCR_MF_STATS
        MF_STAT(9, ck2 - ck1); MF_STAT(10, ck3 - ck2);
#endif
        if (!ok || N == 0) {
            pending = pending || coop;       // (N == 0 cannot happen -- the seed's row is reached -- but costs nothing to cover)
        } else {
            MF_STAT(1, 1); MF_STAT(2, (N + 31u) >> 5); MF_STAT(3, rows);
            // pad the last tile with copies of the first candidate (a duplicate changes no minimum... but it WOULD tie the two
            // smallest group minima: the pad is masked out of the sweep instead, see `valid`)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // ---- coordinates relative to the centre O of the queries' bounding box.  What the MFMA computes for a candidate c'
            // and a query q' is  fl(|q'|^2 + bias) + fl-chain(|c'|^2 - 2 q'.c'):  4 fused multiply-adds whose partial sums are
            // bounded by M^2 = (|q'| + |c'|)^2, |c'|^2 and |q'|^2 (3 roundings each), one addition: |error| < 11 x 2^-24 M^2
            // = 6.6e-7 M^2; kappa = 1e-6 with margin.
            const float big = 3.0e38f;
            const float qx0 = -__ockl_wfred_max_f32(coop ? -tx : -big), qx1 = __ockl_wfred_max_f32(coop ? tx : -big);
            const float qy0 = -__ockl_wfred_max_f32(coop ? -ty : -big), qy1 = __ockl_wfred_max_f32(coop ? ty : -big);
            const float qz0 = -__ockl_wfred_max_f32(coop ? -tz : -big), qz1 = __ockl_wfred_max_f32(coop ? tz : -big);
            const float Ox = 0.5f * (qx0 + qx1), Oy = 0.5f * (qy0 + qy1), Oz = 0.5f * (qz0 + qz1);
            // every candidate lies in the box of cells, every query within Q of O
            const float cxm = fmaxf(Ox - (g.ox + (float)X0 * g.h), g.ox + (float)(X1 + 1) * g.h - Ox) + g.slack;
            const float cym = fmaxf(Oy - (g.oy + (float)Y0 * g.h), g.oy + (float)(Y1 + 1) * g.h - Oy) + g.slack;
            const float czm = fmaxf(Oz - (g.oz + (float)Z0 * g.h), g.oz + (float)(Z1 + 1) * g.h - Oz) + g.slack;
            const float Rc = __builtin_sqrtf(cxm * cxm + cym * cym + czm * czm) * 1.00001f;
            const float hx = 0.5f * (qx1 - qx0), hy = 0.5f * (qy1 - qy0), hz = 0.5f * (qz1 - qz0);
            const float Q = __builtin_sqrtf(hx * hx + hy * hy + hz * hz) * 1.00001f + g.slack;
            const float kappa = 1.0e-6f;
            const float bias = 2.0f * kappa * (Q + Rc) * (Q + Rc);      // keeps every approximate distance positive
            // a lane that does not take part gets a harmless query (O itself): its columns are never read
            const float qx = coop ? tx - Ox : 0.f, qy = coop ? ty - Oy : 0.f, qz = coop ? tz - Oz : 0.f;
            const float px = mf_shfl32(qx), py = mf_shfl32(qy), pz = mf_shfl32(qz);
            // block 0 serves the queries of lanes 0..31, block 1 those of lanes 32..63: in lane l, query (l & 31) of block b
            const float u0x = half ? px : qx, u0y = half ? py : qy, u0z = half ? pz : qz;     // block 0's query of this column
            const float u1x = half ? qx : px, u1y = half ? qy : py, u1z = half ? qz : pz;     // block 1's
            const float n0 = __builtin_fmaf(u0z, u0z, __builtin_fmaf(u0y, u0y, u0x * u0x)) + bias;
            const float n1 = __builtin_fmaf(u1z, u1z, __builtin_fmaf(u1y, u1y, u1x * u1x)) + bias;
            // B[k][j]: k = 0, 1 (first instruction) = x, y; k = 2, 3 (second) = z, the |c'|^2 column
            const float b00 = half ? -2.f * u0y : -2.f * u0x, b01 = half ? 1.f : -2.f * u0z;
            const float b10 = half ? -2.f * u1y : -2.f * u1x, b11 = half ? 1.f : -2.f * u1z;
            MfTrack m0 = {PCR_MF_NONE, PCR_MF_NONE, 0u}, m1 = {PCR_MF_NONE, PCR_MF_NONE, 0u};
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            const uint32_t T = (N + 31u) >> 5;
            // ---- 3b. the sweep, PCR_MF_PIPE tiles' records in flight (a tile is one dependent load + 4 MFMAs: unpipelined, the
            // wave waits a full memory round trip per 32 candidates)
            auto fetch = [&](uint32_t t, PtF &p, bool &valid) {
                const uint32_t k = 32u * t + (uint32_t)(lane & 31);
                valid = k < N;
                p = a.pts[w.cidx[valid ? k : 0u]];
            };
            PtF pq[PCR_MF_PIPE];
            bool vq[PCR_MF_PIPE];
#pragma unroll
            for (int u = 0; u < PCR_MF_PIPE; ++u) fetch((uint32_t)u, pq[u], vq[u]);
            for (uint32_t t0 = 0; t0 < T; t0 += PCR_MF_PIPE) {
#pragma unroll
                for (int u = 0; u < PCR_MF_PIPE; ++u) {
                    const uint32_t t = t0 + (uint32_t)u;
                    const PtF p = pq[u];
                    const bool valid = vq[u];
                    fetch(t + PCR_MF_PIPE, pq[u], vq[u]);                           // (behind the end: candidate 0 again, never used)
                    if (t < T) {
                        const float cx = valid ? p.x - Ox : 3.0e18f, cy = p.y - Oy, cz = p.z - Oz;
                        const float n2 = __builtin_fmaf(cz, cz, __builtin_fmaf(cy, cy, cx * cx));
                        const float a0 = half ? cy : cx, a1 = half ? n2 : cz;
                        f32x16 acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b00, zero, 0, 0, 0);
                        f32x16 acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b10, zero, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b01, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b11, acc1, 0, 0, 0);
                        mf_track(m0, acc0, n0, t);
                        mf_track(m1, acc1, n1, t);
                    }
                }
            }
            MF_CLK(ck4);
            // ---- 4. exact test of the winning group of either block (this lane's half of the rows)
            const float ptx = mf_shfl32(tx), pty = mf_shfl32(ty), ptz = mf_shfl32(tz);
            MfRes res[2];
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                const MfTrack &m = blk ? m1 : m0;
                const bool mine = (blk == 1) == half;                            // this block's query of my column is my own
                const float wx = mine ? tx : ptx, wy = mine ? ty : pty, wz = mine ? tz : ptz;
                float bd = a.bound2_f;
                uint32_t ej = PCR_NONE, eo = PCR_NONE;
                if (m.b1 != PCR_MF_NONE) {
                    const uint32_t kb = 32u * m.t1 + 8u * (m.b1 & 3u) + (half ? 4u : 0u);
                    uint32_t j4[4];
                    PtF p4[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const uint32_t k = kb + (uint32_t)u;
                        j4[u] = w.cidx[k < N ? k : 0u];                          // (a masked pad slot: candidate 0, a real record)
                        j4[u] = j4[u] < n_pad ? j4[u] : n_pad;
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) p4[u] = a.pts[j4[u]];
#pragma unroll
                    for (int u = 0; u < 4; ++u) nn_test<float, PtF, 0>(p4[u], j4[u], wx, wy, wz, bd, ej, eo);
                }
                res[blk].d = bd; res[blk].j = ej; res[blk].o = eo;
                res[blk].s = __uint_as_float(m.b2);                              // (PCR_MF_NONE reads as a NaN: "nothing else", handled below)
            }
            // ---- 5. merge the halves: each lane keeps its own query and receives the other half's result for it
            const MfRes own = half ? res[1] : res[0], snd = half ? res[0] : res[1];
            MfRes rcv;
            rcv.d = mf_shfl32(snd.d); rcv.j = mf_shfl32(snd.j); rcv.o = mf_shfl32(snd.o); rcv.s = mf_shfl32(snd.s);
            {
                const MfRes *cand[2] = {&own, &rcv};
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const unsigned long long key = ((unsigned long long)__float_as_uint(cand[k]->d) << 32) | cand[k]->o;
                    const unsigned long long cur = ((unsigned long long)__float_as_uint(best) << 32) | bo;
                    const bool take = cand[k]->o != PCR_NONE && key < cur;
                    best = take ? cand[k]->d : best; bj = take ? cand[k]->j : bj; bo = take ? cand[k]->o : bo;
                }
            }
            // Certificate.  A candidate that was not tested exactly has an approximate distance >= s (the smaller of the two halves'
            // second group minima; packing cleared / set two low bits: 1e-6 of it).  If it lies farther than rho0 = |q'| + sqrt(best)
            // from O it is farther than sqrt(best) from the query by the triangle inequality; if not, M <= 2 |q'| + sqrt(best) bounds
            // its error, and its true distance is >= s - bias - kappa M^2.  Strictly above `best`: the winner is unique.
            const uint32_t so = __float_as_uint(own.s), sr = __float_as_uint(rcv.s);
            const float s_own = so >= 0x7f800000u ? RT::inf() : own.s, s_rcv = sr >= 0x7f800000u ? RT::inf() : rcv.s;
            const float smin = fminf(s_own, s_rcv);
            const float qn = __builtin_sqrtf(__builtin_fmaf(qz, qz, __builtin_fmaf(qy, qy, qx * qx))) * 1.00001f;
            const float Mn = 2.0f * qn + __builtin_sqrtf(best) * 1.00001f + g.slack;
            const float lb = smin < RT::inf() ? smin * 0.999999f - bias - kappa * Mn * Mn : smin;
            const bool cert = lb > best * 1.000001f;
            MF_STAT(6, __popcll(__ballot(coop && !cert)));
            MF_CLK(ck5);
#ifdef PCR_MF_STATS
            MF_STAT(11, ck4 - ck3); MF_STAT(12, ck5 - ck4);
#endif
            pending = pending || (coop && !cert);
        }
    }
    // ---- 6. whatever could not be settled goes to the wave's list (searched 64 at a time: nn_mfma_flush)
    const unsigned long long pm = __ballot(pending);
    if (pm != 0ull) {
        const int np = __popcll(pm);
        if (w.ndefer + np > 64) { MF_STAT(7, 1); nn_mfma_flush<HALO>(a, g, P, w); }
        if (pending) w.defer[w.ndefer + __popcll(pm & ((1ull << lane) - 1ull))] = (uint32_t)i;
        w.ndefer += np;
    }
    if (exists && !pending) {
        const bool okm = live && bo != PCR_NONE && __builtin_sqrtf(best) < a.md_f;
        a.nn_j[i] = okm ? bj : PCR_NONE;
    }
    MF_CLK(ck6);
#ifdef PCR_MF_STATS
    MF_STAT(14, ck6 - ck0);
#endif
}
