// Device-side building blocks shared by the shipped kernels (kernels.hip) and the developer / A-B kernels
// (kernels_dev.hip): pass arguments, pose handling, per-correspondence accumulation, the folds, tile hand-out,
// the per-lane search of one scan point.
#pragma once

#include <string.h>
#include <time.h>

#include "gn_math.h"
#include "nn_device.h"

// what the kernels need of a pose: float32 copy for the point transform, float64 rotation for the Jacobians
struct PoseK {
    float r32[9], t32[3];
    double R[9];
};
// the transform half of a pose only (the previous pass' pose)
struct PoseQ {
    float r32[9], t32[3];
};

struct LinArgs {
    // scan
    const float *sx, *sy, *sz;
    int64_t n;
    // point target
    Geom<float> gf;
    const PtF *pts;
    uint32_t pts_last;               // index of the last readable record of pts (its sentinels included)
    const PtN *pn;
    // voxel target
    Geom<double> gd;
    const PtD *means;
    const double *vnorm;
    const double *vicov;
    const uint32_t *cell_start;
    // pose: by value (pcr_linearize: the caller's T) or, when `pose` is set, read from HBM at kernel
    // start (pcr_align: the device-resident Gauss-Newton loop; pose->done != 0 turns the launch into a no-op)
    PoseK hp;
    const PoseDev *pose;
    // certified reuse: the float32 pose of the previous pass over this scan (k_certify, tracking searches)
    PoseQ hq;
    float *lb2;                      // per scan point: lower bound on the distance to every target point but its match
    unsigned long long *umask;       // per 64-point tile: lanes k_certify could not certify
    uint32_t *ucnt;                  // per k_certify block: points marked
    float mu_f;                      // margin of a tracking search (metres); also the per-point motion gate
    float md_f;        // gate, float32 compare (point targets)
    double md_d;       // gate, float64 compare (voxel targets)
    float bound2_f;    // search bound (squared), slightly above the gate
    double bound2_d;
    // float32 filter of a plain centroid search (nn_point_filter; gf / pts then describe the index over the ROUNDED centroids)
    const uint32_t *cs_f;   // its cell_start
    float band_f;           // how far rounding to float32 may move a centroid (metres); 0 = no filter
    float mu_ff;            // tracking margin of the filter search: 2 band + 3e-5 of the bound
    float bound2_ff;        // its search bound (squared): the float64 bound + band, a little more
    uint32_t *pending;      // one word: the stamp of the last filter pass that left PCR_PENDING entries in nn_j (k_nn_fix)
    uint32_t stamp;         // this pass' stamp (increasing per context)
    unsigned flags;
    int nblocks;
    double *partials;  // [nblocks + 8][32]
    // variant 1: correspondences through HBM
    uint32_t *nn_j;
    uint32_t *tile_ctr;   // tile counters (64 B apart) of the NN kernels' dynamic hand-out
    int sched_local;      // 1: block-local hand-out (small scans), 0: global counters (see nn_tile_loop)
    // the deeper set of extended lists of a point target (pcr_target::cs_h2 ...; halo2_f = 0: none).  Host-driven passes
    // put the set they chose into gf; the device-resident loop (pose != NULL) picks per iteration (PoseDev::halo_deep)
    const uint32_t *cs_h2;
    const void *pts_h2;
    const uint32_t *j_h2;
    float halo2_f;
    const float4 *lbox_h2, *gbox_h2;     // leaf / group boxes of the deeper lists (heavy targets)
};

// the geometry a point search of this launch reads: gf, with the deeper lists swapped in when the device-resident loop
// asked for them (wave-uniform: a handful of scalar selects at kernel start)
__device__ __forceinline__ Geom<float> select_lists(const LinArgs &a) {
    Geom<float> g = a.gf;
    if (a.pose != nullptr && a.halo2_f > 0.f && __builtin_amdgcn_readfirstlane(a.pose->halo_deep) != 0) {
        g.halo = a.halo2_f; g.cs_h = a.cs_h2; g.pts_h = a.pts_h2; g.j_h = a.j_h2; g.lbox_h = a.lbox_h2; g.gbox_h = a.gbox_h2;
    }
    return g;
}

__device__ __forceinline__ float uniform_f32(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
__device__ __forceinline__ double uniform_f64(double v) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

// The pose of this launch, in scalar registers.  Returns false when the device-resident loop has
// already finished (nothing to do).  Every block reads the pose at its start; the pose is rewritten
// only by k_gn_update, a separate launch on the same stream behind the reduce kernel, so a read never
// races with the update: the kernel boundary orders them.
template <bool NEED_R>
__device__ __forceinline__ bool load_pose(const LinArgs &a, PoseK &P) {
    if (a.pose == nullptr) { P = a.hp; return true; }
    const PoseDev *p = a.pose;
    if (__builtin_amdgcn_readfirstlane(p->done) != 0) return false;
#pragma unroll
    for (int i = 0; i < 9; ++i) P.r32[i] = uniform_f32(p->r32[i]);
#pragma unroll
    for (int i = 0; i < 3; ++i) P.t32[i] = uniform_f32(p->t32[i]);
    if (NEED_R) {
#pragma unroll
        for (int i = 0; i < 9; ++i) P.R[i] = uniform_f64(p->R[i]);
    }
    return true;
}

// the previous pass' float32 pose (certified reuse runs on host-driven passes only: by value)
__device__ __forceinline__ void load_prev(const LinArgs &a, PoseQ &Q) { Q = a.hq; }

template <typename POSE>
__device__ __forceinline__ void xform(const POSE &a, float x, float y, float z, float &tx, float &ty, float &tz) {
    // ((R00*x + R01*y) + R02*z) + t0, float32, no contraction: oracle orc_transform
    tx = ((a.r32[0] * x + a.r32[1] * y) + a.r32[2] * z) + a.t32[0];
    ty = ((a.r32[3] * x + a.r32[4] * y) + a.r32[5] * z) + a.t32[1];
    tz = ((a.r32[6] * x + a.r32[7] * y) + a.r32[8] * z) + a.t32[2];
}

// ---- per-correspondence accumulation ------------------------------------------------------
// acc layout for PLANE / VPLANE / NDT: 0..20 triu(H), 21..26 g, 27 e2, 28 count.
// acc layout for ICP (closed form, icp.py:40-47): 0 count, 1..3 sum p, 4..9 second moments
// (xx xy xz yy yz zz), 10..12 sum r, 13..15 sum p x v (v = R r or R^T r), 16 e2.

__device__ __forceinline__ void acc_rank1(double *acc, const double J[6], double r) {
    int p = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = i; j < 6; ++j) { acc[p] = fma(J[i], J[j], acc[p]); ++p; }
#pragma unroll
    for (int i = 0; i < 6; ++i) acc[21 + i] = fma(J[i], r, acc[21 + i]);
    acc[27] = fma(r, r, acc[27]);
    acc[28] += 1.0;
}

__device__ __forceinline__ void acc_plane(double *acc, const PoseK &a, double x, double y, double z,
                                          double n0, double n1, double n2, double d0, double d1, double d2) {
    const double r = (n0 * d0 + n1 * d1) + n2 * d2;                          // plane_icp.py:49
    const double ra = a.R[0] * n0 + a.R[3] * n1 + a.R[6] * n2;               // R^T n, plane_icp.py:51
    const double rb = a.R[1] * n0 + a.R[4] * n1 + a.R[7] * n2;
    const double rc = a.R[2] * n0 + a.R[5] * n1 + a.R[8] * n2;
    const double J[6] = {n0, n1, n2, -z * rb + y * rc, z * ra - x * rc, -y * ra + x * rb};   // math_tools.py:22-31
    acc_rank1(acc, J, r);
}

__device__ __forceinline__ void acc_icp(double *acc, const PoseK &a, unsigned flags, double x, double y, double z,
                                        double r0, double r1, double r2) {
    acc[0] += 1.0;
    acc[1] += x; acc[2] += y; acc[3] += z;
    acc[4] = fma(x, x, acc[4]); acc[5] = fma(x, y, acc[5]); acc[6] = fma(x, z, acc[6]);
    acc[7] = fma(y, y, acc[7]); acc[8] = fma(y, z, acc[8]); acc[9] = fma(z, z, acc[9]);
    acc[10] += r0; acc[11] += r1; acc[12] += r2;
    double v0, v1, v2;
    if (flags & PCR_FLAG_ICP_RR_QUIRK) {                                     // quirk Q1, icp.py:53-54
        v0 = a.R[0] * r0 + a.R[1] * r1 + a.R[2] * r2;
        v1 = a.R[3] * r0 + a.R[4] * r1 + a.R[5] * r2;
        v2 = a.R[6] * r0 + a.R[7] * r1 + a.R[8] * r2;
    } else {                                                                 // consistent J^T r, icp.py:81-87
        v0 = a.R[0] * r0 + a.R[3] * r1 + a.R[6] * r2;
        v1 = a.R[1] * r0 + a.R[4] * r1 + a.R[7] * r2;
        v2 = a.R[2] * r0 + a.R[5] * r1 + a.R[8] * r2;
    }
    acc[13] += y * v2 - z * v1; acc[14] += z * v0 - x * v2; acc[15] += x * v1 - y * v0;
    acc[16] += r0 * r0 + r1 * r1 + r2 * r2;
}

__device__ __forceinline__ void acc_ndt(double *acc, const PoseK &a, double x, double y, double z,
                                        const double *__restrict__ c6, double d0, double d1, double d2) {
    // J = [I, A] with A = -R skew(p) (ndt.py:40); C symmetric inverse covariance.  The identity block is exploited by
    // hand: H_ll = C, H_lr = C A, H_rr = A^T (C A), g = [C d, A^T (C d)] -- the generic J^T C J spends 60 % of its
    // multiplications on exact ones and zeros, which the compiler may not drop (0 * x is not 0 for a NaN).  Every
    // non-trivial entry is evaluated in the order the generic form used ((p0 + p1) + p2), so the sums are bit-identical
    // to round 2's for finite inputs.
    const double C[3][3] = {{c6[0], c6[1], c6[2]}, {c6[1], c6[3], c6[4]}, {c6[2], c6[4], c6[5]}};
    double A[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double ri0 = a.R[3 * i], ri1 = a.R[3 * i + 1], ri2 = a.R[3 * i + 2];
        // -(R S) with S = [[0,-z,y],[z,0,-x],[-y,x,0]]
        A[i][0] = -(ri1 * z - ri2 * y);
        A[i][1] = -(-ri0 * z + ri2 * x);
        A[i][2] = -(ri0 * y - ri1 * x);
    }
    double CA[3][3], Cd[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        Cd[i] = C[i][0] * d0 + C[i][1] * d1 + C[i][2] * d2;
#pragma unroll
        for (int j = 0; j < 3; ++j) CA[i][j] = C[i][0] * A[0][j] + C[i][1] * A[1][j] + C[i][2] * A[2][j];
    }
    // upper triangle, row-major: rows 0..2 = [C (upper part) | C A], rows 3..5 = A^T (C A) (upper part)
    int p = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = i; j < 3; ++j) { acc[p] += C[i][j]; ++p; }
#pragma unroll
        for (int j = 0; j < 3; ++j) { acc[p] += CA[i][j]; ++p; }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = i; j < 3; ++j) { acc[p] += A[0][i] * CA[0][j] + A[1][i] * CA[1][j] + A[2][i] * CA[2][j]; ++p; }
#pragma unroll
    for (int i = 0; i < 3; ++i) acc[21 + i] += Cd[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) acc[24 + i] += A[0][i] * Cd[0] + A[1][i] * Cd[1] + A[2][i] * Cd[2];
    acc[27] += d0 * Cd[0] + d1 * Cd[1] + d2 * Cd[2];
    acc[28] += 1.0;
}

// the gate of the reference (icp.py:34, plane_icp.py:41, voxelized_plane_icp.py:38, ndt.py:33: dist < max_dist, strict), on
// the distance exactly as the search computes it (nn_test): the reduce kernels apply it themselves, so that the
// search may leave UNGATED matches behind for the next pass (certified reuse)
// (PCR_IFLAG_NOGATE, quirk Q6: the float64 search of a float64 point target already applied the reference's float64 gate
// -- its tree returns float64 distances, plane_icp.py:22,40-41 -- so the float32 records' own distance must not gate again)
#define PCR_IFLAG_NOGATE (1u << 27)
__device__ __forceinline__ bool gate_f32(const LinArgs &a, float dx, float dy, float dz) {
    return (a.flags & PCR_IFLAG_NOGATE) != 0 || __builtin_sqrtf(dist2_f32(dx, dy, dz)) < a.md_f;
}
__device__ __forceinline__ bool gate_f64(const LinArgs &a, double dx, double dy, double dz) {
    return __builtin_sqrt((dx * dx + dy * dy) + dz * dz) < a.md_d;
}

// gather the matched record at cell-sorted index j and accumulate (GATE: apply the distance gate here)
template <int KIND, bool GATE>
__device__ __forceinline__ void accumulate(double *acc, const LinArgs &a, const PoseK &P, uint32_t j,
                                           float x, float y, float z, float tx, float ty, float tz) {
    if (KIND == PCR_ICP) {
        const PtF q = a.pts[j];
        const float dx = tx - q.x, dy = ty - q.y, dz = tz - q.z;
        if (GATE && !gate_f32(a, dx, dy, dz)) return;
        acc_icp(acc, P, a.flags, x, y, z, (double)dx, (double)dy, (double)dz);   // icp.py:39
    } else if (KIND == PCR_PLANE) {
        // point and normal from ONE 32-byte record (two 16-byte loads of the same sector)
        const float4 *rec = reinterpret_cast<const float4 *>(a.pn + j);
        const float4 q = rec[0], nn = rec[1];
        const float dx = tx - q.x, dy = ty - q.y, dz = tz - q.z;
        if (GATE && !gate_f32(a, dx, dy, dz)) return;
        acc_plane(acc, P, x, y, z, nn.x, nn.y, nn.z, (double)dx, (double)dy, (double)dz);
    } else if (KIND == PCR_VPLANE) {
        const PtD q = a.means[j];
        const double dx = (double)tx - q.x, dy = (double)ty - q.y, dz = (double)tz - q.z;
        if (GATE && !gate_f64(a, dx, dy, dz)) return;
        const double *nn = a.vnorm + 3 * (size_t)j;
        acc_plane(acc, P, x, y, z, nn[0], nn[1], nn[2], dx, dy, dz);
    } else {
        const PtD q = a.means[j];
        const double dx = (double)tx - q.x, dy = (double)ty - q.y, dz = (double)tz - q.z;
        if (GATE && !gate_f64(a, dx, dy, dz)) return;
        acc_ndt(acc, P, x, y, z, a.vicov + 6 * (size_t)j, dx, dy, dz);
    }
}

// The streaming loop of the reduce kernels.  Point targets: the matched records of TWO scan points are
// gathered before either is accumulated (two independent 16/32-byte gathers in flight per lane: at
// 1e8 target points every gather is an HBM miss and the kernel is bound by misses in flight); the
// points are still accumulated in index order, so the sums are bit-identical to the one-at-a-time loop.
// Round 5, measured and NOT kept: FOUR CONSECUTIVE scan points per lane (index and coordinates as 16-byte
// loads, four gathers in flight) is slower -- 24.9 vs 20.8 us at 1.06 M points, 150 vs 117 / 351 vs 270 us at 1e8 -- because a
// wave-level gather then covers every 4th of 256 points instead of 64 neighbours: ~3x the distinct lines per instruction, and
// the L1 tag rate is what gathers pay with (docs/EXPERIMENTS.md).  PCR_RED_W selects how many scan points' gathers a lane has in
// flight in the neighbour-preserving mapping below (lane l of a tile = point l of 64 consecutive ones).
#ifndef PCR_RED_W
#define PCR_RED_W 2
#endif
#ifndef PCR_RED_NT
#define PCR_RED_NT 0
#endif
template <typename T>
__device__ __forceinline__ T stream_load(const T *p) {
#if PCR_RED_NT
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}

template <int KIND>
__device__ __forceinline__ void reduce_stream(double *acc, const LinArgs &a, const PoseK &P, int64_t base, int64_t end,
                                              int64_t stride) {
    if (KIND == PCR_ICP || KIND == PCR_PLANE) {
        constexpr int W = PCR_RED_W;
        for (int64_t i = base; i < end; i += W * stride) {
            uint32_t j[W];
            float4 q[W], nr[W];
#pragma unroll
            for (int u = 0; u < W; ++u) {
                const int64_t iu = i + u * stride;
                j[u] = iu < end ? stream_load(a.nn_j + iu) : PCR_NONE;
            }
#pragma unroll
            for (int u = 0; u < W; ++u) {
                q[u] = make_float4(0, 0, 0, 0); nr[u] = q[u];
                if (j[u] != PCR_NONE) {
                    if (KIND == PCR_PLANE) { const float4 *r = reinterpret_cast<const float4 *>(a.pn + j[u]); q[u] = r[0]; nr[u] = r[1]; }
                    else q[u] = a.pts[j[u]];
                }
            }
#pragma unroll
            for (int u = 0; u < W; ++u) {
                if (j[u] == PCR_NONE) continue;
                const int64_t iu = i + u * stride;
                const float x = stream_load(a.sx + iu), y = stream_load(a.sy + iu), z = stream_load(a.sz + iu);
                float tx, ty, tz;
                xform(P, x, y, z, tx, ty, tz);
                const float dx = tx - q[u].x, dy = ty - q[u].y, dz = tz - q[u].z;
                if (gate_f32(a, dx, dy, dz)) {
                    if (KIND == PCR_PLANE) acc_plane(acc, P, x, y, z, nr[u].x, nr[u].y, nr[u].z, (double)dx, (double)dy, (double)dz);
                    else acc_icp(acc, P, a.flags, x, y, z, (double)dx, (double)dy, (double)dz);
                }
            }
        }
    } else {
        for (int64_t i = base; i < end; i += stride) {
            const uint32_t j = a.nn_j[i];
            if (j == PCR_NONE) continue;
            const float x = a.sx[i], y = a.sy[i], z = a.sz[i];
            float tx, ty, tz;
            xform(P, x, y, z, tx, ty, tz);
            accumulate<KIND, true>(acc, a, P, j, x, y, z, tx, ty, tz);
        }
    }
}

// ---- block reduction of 32 float64 sums --------------------------------------------------
__device__ __forceinline__ double shfl_xor_f64(double v, int mask) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, mask, 64);
    hi = __shfl_xor(hi, mask, 64);
    return __hiloint2double(hi, lo);
}

// One halving step: lanes whose `MASK` bit is clear keep components [0, HALF), the others keep
// [HALF, 2*HALF); each lane adds its partner's copy of what it keeps.  HALF and MASK are template
// constants so every acc[] index is static (a runtime-indexed array would live in scratch).
template <int HALF, int MASK>
__device__ __forceinline__ void fold_step(double *acc, int lane) {
    const bool upper = (lane & MASK) != 0;
#pragma unroll
    for (int i = 0; i < HALF; ++i) {
        const double send = upper ? acc[i] : acc[i + HALF];
        const double keep = upper ? acc[i + HALF] : acc[i];
        acc[i] = keep + shfl_xor_f64(send, MASK);
    }
}

// After the call lane l holds the wave-wide sum of component (l >> 1) in acc[0].
// SPLIT (k_scan_reduce): the lanes come in holding 16 values each -- components 0..15 on lanes 0..31, 16..31 on lanes 32..63 --
// i.e. the state after the first halving step.
template <bool SPLIT = false>
__device__ __forceinline__ void wave_fold32(double *acc, int lane) {
    if (!SPLIT) fold_step<16, 32>(acc, lane);
    fold_step<8, 16>(acc, lane);
    fold_step<4, 8>(acc, lane);
    fold_step<2, 4>(acc, lane);
    fold_step<1, 2>(acc, lane);
    acc[0] += shfl_xor_f64(acc[0], 1);
}

// COHERENT: the store is written through to memory at agent scope, so that a block on ANOTHER XCD
// (each XCD has a private, mutually non-coherent L2) can read it inside the same kernel.
template <bool COHERENT = false, bool SPLIT = false>
__device__ __forceinline__ void block_store_partials(double *acc, double *__restrict__ partials) {
    __shared__ double wsum[4][32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    wave_fold32<SPLIT>(acc, lane);
    if ((lane & 1) == 0) wsum[wave][lane >> 1] = acc[0];
    __syncthreads();
    if (threadIdx.x < 32) {
        const double s = ((wsum[0][threadIdx.x] + wsum[1][threadIdx.x]) + wsum[2][threadIdx.x]) + wsum[3][threadIdx.x];
        double *dst = &partials[(size_t)blockIdx.x * 32 + threadIdx.x];
        if (COHERENT) __hip_atomic_store(dst, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else *dst = s;
    }
}

// Work distribution: the sorted scan is cut into 8 contiguous spans, one per XCD (block b runs on
// XCD b % 8, so each XCD's private L2 serves one region of space); inside a span the 256-point
// tiles are dealt round-robin to that XCD's blocks, which evens out regions where the search is
// slow (large residual offsets) without giving up the locality.
struct TileIter {
    int64_t base, end, stride;
    __device__ __forceinline__ TileIter(const LinArgs &a) {
        const int per = a.nblocks >> 3;                         // blocks per XCD
        const int xcd = (int)(blockIdx.x & 7), bi = (int)(blockIdx.x >> 3);
        const int64_t span = (((a.n + 7) >> 3) + 255) & ~(int64_t)255;
        const int64_t lo = span * xcd;
        end = lo + span < a.n ? lo + span : a.n;
        base = lo + (int64_t)bi * 256 + threadIdx.x;
        stride = (int64_t)per * 256;
    }
};

// ---- fused form: everything in one kernel ------------------------------------------------------
// Slower than search + reduce for large scans (107-157 VGPRs: half the occupancy of k_nn_scan) but FASTER
// for small ones, where a pass is a chain of dependent cold misses rather than a throughput problem: one
// launch less, no round trip of the matches through HBM (100 k-point scan: 48.8 vs 58.9 us per pass).
// The host picks per launch (pcr_set_variant: 2 = automatic, the default).
template <int HALO, int SETTLE, int B, int Q6 = 0>
__device__ __forceinline__ bool nn_filter_core(const LinArgs &a, float tx, float ty, float tz, uint32_t &w, double &d);

// FILT (voxel kinds, round 4): the centroid search runs the float32 filter search with two-way settling (nn_filter_core)
// and falls back to the float64 search inline for what that cannot certify (a third centroid inside the margin: rare
// enough that the extra chain does not show); HALO then says whether the FILTER index has the extended lists.
template <int KIND, int HALO, int FILT, int LB = 0>
__device__ __forceinline__ void linearize_body(const LinArgs &a, const PoseK &P, double *acc) {
    const TileIter it(a);
    for (int64_t i = it.base; i < it.end; i += it.stride) {
        const float x = a.sx[i], y = a.sy[i], z = a.sz[i];
        float tx, ty, tz;
        xform(P, x, y, z, tx, ty, tz);
        uint32_t bj, bo;
        bool ok;
        if (KIND == PCR_ICP || KIND == PCR_PLANE) {
            float best;
            nn_search<float, PtF, false, false, HALO != 0, false, false, PCR_NN_BATCH_SMALL, false, LB != 0>(a.gf, a.pts, a.cell_start, tx, ty, tz,
                                                                                                          a.bound2_f, best, bj, bo);
            ok = bj != PCR_NONE && __builtin_sqrtf(best) < a.md_f;                 // icp.py:34 strict gate
        } else if (FILT) {
            double d;
            bool cert = nn_filter_core<HALO, 1, PCR_NN_BATCH_SMALL>(a, tx, ty, tz, bj, d);
            if (!cert) {
                nn_search<double, PtD, false, false, false, false, false, PCR_NN_BATCH_SMALL>(a.gd, a.means, a.cell_start, (double)tx, (double)ty,
                                                                                           (double)tz, a.bound2_d, d, bj, bo);
            }
            ok = bj != PCR_NONE && __builtin_sqrt(d) < a.md_d;                    // voxelized_plane_icp.py:38
        } else {
            double best;
            nn_search<double, PtD, false, false, false, false, false, PCR_NN_BATCH_SMALL>(a.gd, a.means, a.cell_start, (double)tx, (double)ty,
                                                                                       (double)tz, a.bound2_d, best, bj, bo);
            ok = bj != PCR_NONE && __builtin_sqrt(best) < a.md_d;                 // voxelized_plane_icp.py:38
        }
        if (ok) accumulate<KIND, false>(acc, a, P, bj, x, y, z, tx, ty, tz);
    }
}

// ---- variant 1: NN kernel (few registers, high occupancy) + streaming reduce kernel ---------
// The cost of a query varies by more than 10x with its distance to the surface, so waves pull
// tiles from counters instead of owning a fixed share: every wave stays busy until its XCD's span of
// the scan is exhausted (the finalize step re-zeroes the counters).
//
// Tile hand-out.  The sorted scan is cut into PCR_TILE_CTRS contiguous sub-spans; sub-spans c, c + 8,
// c + 16, ... belong to XCD c & 7 (blocks b with b % 8 == c run there: a locality assumption only).
// A wave's first PCR_TILE_STATIC_ROUNDS tiles of its home sub-span are fixed by its index (no atomic:
// thousands of waves asking the same word at launch serialise at ~30 ns each), the rest of every
// sub-span is handed out by a counter; a wave that finds its home sub-span empty moves on to the
// other sub-spans of its XCD.  Counters live PCR_TILE_STRIDE words apart.
#ifndef PCR_TILE_CTRS
#define PCR_TILE_CTRS 64       // measured on MI355X (1.06 M queries, skeleton without the search): 8 counters
#endif                         // and no static round 64 us, 8 + static 45, 64: 34, 64 + static 31, no counters 7
#ifndef PCR_TILE_STRIDE
#define PCR_TILE_STRIDE 32
#endif
#ifndef PCR_TICKET_STRIDE
#define PCR_TICKET_STRIDE 32         // words between the tickets of k_reduce_finalize's fold: a 128-byte line each
#endif
#ifndef PCR_TILE_STATIC_ROUNDS
#define PCR_TILE_STATIC_ROUNDS 1   // 2 static rounds already unbalance the far poses (whole kernel 127 -> 155 us)
#endif
#define PCR_TILE_SUB (PCR_TILE_CTRS / 8)
// calls body(first, end) wave-uniformly for every TP-point tile this wave is given (TP = 64: lane l owns scan
// point first + l, which exists iff first + l < end; TP = 1024: a chunk of a LIST pass, see nn_chunk_list)
// Two hand-out policies, chosen per launch (LinArgs::sched_local):
//  * block-local (mid-size scans: at most ~1.5 tiles per launched wave, i.e. up to ~590 k points; below
//    ~262 k the fused kernel runs instead): the XCD's span is dealt round-robin to the XCD's blocks (block
//    b owns tiles b, b + B, ...) and a block's four waves pull from that list through ONE counter in LDS --
//    no global atomics (they alone cost 24 us of a 1.06 M-point pass).  Measured per pass: 300 k points
//    91.7 vs 104.8 us, 450 k 105.5 vs 118.8.
//  * global counters (everything larger): PCR_TILE_CTRS sub-spans, one static round, then device-wide
//    counters.  With many tiles per wave and costs that differ 10x between regions the static deal
//    loses more than the atomics cost (1.06 M: 134 vs 147 us; 1e8-point target: 3.3 vs 4.9 ms).
// Chunk interleave (round 5; PCR_TILE_INTERLEAVE, compile-time since both schemes in one kernel spill).  The schemes below give XCD x the contiguous eighth x of the sorted scan
// (LOCAL) or eight of 64 contiguous sub-spans (counters): good for the XCD's L2, bad when the COST is not spread like the points --
// a scan that overlaps the map only partly (bench config plane_b01_crop: 70 % of the points leave the search at once) keeps two or
// three XCDs busy and five idle (search 200-220 us at every pose where the work is worth ~60).  With the interleave the positions a
// scheme hands out are read as VIRTUAL: virtual tile v of XCD x is the real tile ((v / CH) * 8 + x) * CH + v % CH, i.e. chunks of
// CH = 16 tiles (1024 points, still one compact patch) dealt round-robin to the XCDs.
#ifndef PCR_TILE_CHUNK
#define PCR_TILE_CHUNK 16
#endif
#ifndef PCR_TILE_INTERLEAVE
#define PCR_TILE_INTERLEAVE 1
#endif
template <int TP>
__device__ __forceinline__ int64_t nn_tile_real(const LinArgs &a, int xcd, int64_t lo_x, int64_t first) {
    // `first` is a position inside XCD xcd's virtual range, which starts at lo_x (both multiples of TP)
    const int64_t v = (first - lo_x) / TP;
    return (((v / PCR_TILE_CHUNK) * 8 + xcd) * PCR_TILE_CHUNK + v % PCR_TILE_CHUNK) * TP;
}

template <int LOCAL, int TP, typename Body>
__device__ __forceinline__ void nn_tile_loop(const LinArgs &a, Body &&body) {
    const int xcd = (int)(blockIdx.x & 7);
    const int lane = threadIdx.x & 63;
    const uint32_t xb = blockIdx.x >> 3, nxb = gridDim.x >> 3;                 // block index / blocks on this XCD
    __shared__ uint32_t blk_next;
    if (LOCAL) {
        if (threadIdx.x == 0) blk_next = 0;
        __syncthreads();
    }
#if PCR_TILE_INTERLEAVE
    {
        // the XCD's virtual range: whole chunks, enough of them for an eighth of the scan's chunks (rounded up)
        const int64_t chunk = (int64_t)PCR_TILE_CHUNK * TP;
        const int64_t nchunks = (a.n + chunk - 1) / chunk;
        const int64_t vspan = ((nchunks + 7) / 8) * chunk;                     // virtual points per XCD
        if (LOCAL) {
            for (;;) {
                uint32_t k = 0;
                if (lane == 0) k = atomicAdd(&blk_next, 1u);
                k = __builtin_amdgcn_readfirstlane(k);
                const int64_t vfirst = ((int64_t)xb + (int64_t)k * nxb) * TP;
                if (vfirst >= vspan) break;
                const int64_t first = nn_tile_real<TP>(a, xcd, 0, vfirst);
                if (first >= a.n) break;                                       // (real positions grow with the virtual ones)
                body(first, first + TP < a.n ? first + TP : a.n);
            }
        } else {
            // counters: the XCD's virtual range cut into PCR_TILE_SUB sub-spans, counter (xcd + 8 sub) each; as below, a wave's
            // first tile of its HOME sub-span is fixed by its index (no atomic), the rest is handed out by the counter
            const int64_t sspan = (((vspan / TP + PCR_TILE_SUB - 1) / PCR_TILE_SUB)) * TP;
            const int home = (int)(xb % PCR_TILE_SUB);
            const uint32_t wrank = (xb / PCR_TILE_SUB) * 4 + (threadIdx.x >> 6);
            for (int r = 0; r < PCR_TILE_SUB; ++r) {
                const int sub = (home + r) % PCR_TILE_SUB;
                const int64_t slo = sspan * sub;
                const int64_t send = slo + sspan < vspan ? slo + sspan : vspan;
                const uint32_t nstatic = ((nxb - sub + PCR_TILE_SUB - 1) / PCR_TILE_SUB) * 4;     // home waves of this sub-span
                bool stat = r == 0;
                for (;;) {
                    uint32_t t = 0;
                    if (stat) { t = wrank; stat = false; }
                    else {
                        if (lane == 0) t = atomicAdd(&a.tile_ctr[(xcd + 8 * sub) * PCR_TILE_STRIDE], 1u);
                        t = __builtin_amdgcn_readfirstlane(t) + nstatic;
                    }
                    const int64_t vfirst = slo + (int64_t)t * TP;
                    if (vfirst >= send) break;
                    const int64_t first = nn_tile_real<TP>(a, xcd, 0, vfirst);
                    if (first >= a.n) break;
                    body(first, first + TP < a.n ? first + TP : a.n);
                }
            }
        }
    }
#else
    // (the round-2 to round-4 scheme: one contiguous eighth / eight contiguous 1/64 sub-spans per XCD; kept as a compile-time
    // variant -- both schemes behind a run-time switch put 784 bytes of k_nn_filter<.., LOCAL = 1>'s state in scratch)
    // global-counter state
    const int64_t gspan = (((a.n + PCR_TILE_CTRS - 1) / PCR_TILE_CTRS) + (TP - 1)) & ~(int64_t)(TP - 1);
    const int home = (int)(xb % PCR_TILE_SUB);
    const uint32_t wrank = (xb / PCR_TILE_SUB) * 4 + (threadIdx.x >> 6);
    const uint32_t wcount = ((nxb - home + PCR_TILE_SUB - 1) / PCR_TILE_SUB) * 4;
    int r = 0, sr = 0;
    // block-local state
    const int64_t lspan = (((a.n + 7) >> 3) + (TP - 1)) & ~(int64_t)(TP - 1);
    for (;;) {
        int64_t first, end;
        if (LOCAL) {
            const int64_t lo = lspan * xcd;
            end = lo + lspan < a.n ? lo + lspan : a.n;
            uint32_t k = 0;
            if (lane == 0) k = atomicAdd(&blk_next, 1u);               // ds_add_rtn_u32: no memory traffic
            k = __builtin_amdgcn_readfirstlane(k);
            first = lo + ((int64_t)xb + (int64_t)k * nxb) * TP;
            if (first >= end) break;
        } else {
            bool got = false;
            for (; r < PCR_TILE_SUB; ++r, sr = PCR_TILE_STATIC_ROUNDS) {     // static rounds only at home (r == 0)
                const int sub = (home + r) % PCR_TILE_SUB;
                const int c = xcd + 8 * sub;
                const int64_t lo = gspan * c;
                end = lo + gspan < a.n ? lo + gspan : a.n;
                // static tiles of sub-span `sub`: PCR_TILE_STATIC_ROUNDS per home wave of that sub-span
                const uint32_t hcount = ((nxb - sub + PCR_TILE_SUB - 1) / PCR_TILE_SUB) * 4;
                const uint32_t nstatic = PCR_TILE_STATIC_ROUNDS * hcount;
                const uint32_t ntiles = end > lo ? (uint32_t)((end - lo + (TP - 1)) / TP) : 0u;
                uint32_t t;
                if (sr < PCR_TILE_STATIC_ROUNDS) {
                    t = wrank + (uint32_t)sr * wcount;
                    ++sr;
                } else {
                    if (nstatic >= ntiles) continue;                   // every tile of this sub-span was a static one
                    t = 0;
                    if (lane == 0) t = atomicAdd(&a.tile_ctr[c * PCR_TILE_STRIDE], 1u);
                    t = __builtin_amdgcn_readfirstlane(t) + nstatic;
                }
                first = lo + (int64_t)t * TP;
                if (first < end) { got = true; break; }
            }
            if (!got) break;
        }
        body(first, end);
    }
#endif
}

// The match of a plain pass goes to HBM for the reduce kernel, which reads it from another XCD more often than not.  Round 6
// tried a NON-TEMPORAL store here (-DPCR_NNJ_NT=1: global_store_dword ... nt), so that the 4 bytes per scan point would not sit
// dirty in the writing XCD's L2 until the end-of-kernel release (40 MB on a 10 M-point scan): no gain on any config, the reduce
// kernel of ndt_10m 5 % slower (its index stream then comes from HBM); profiles/r06_nt_ab.txt.  Plain stores stay.
#ifndef PCR_NNJ_NT
#define PCR_NNJ_NT 0
#endif
__device__ __forceinline__ void nnj_store(uint32_t *p, uint32_t v) {
#if PCR_NNJ_NT
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}

// One query: scan point i, on its own lane (gathers): the general search.  HALO: the target has the extended
// per-cell lists and ring 0 reads those (nn_ring0).
// TRACK = 0: the match is gated here (PCR_NONE = no correspondence) -- nothing else is left behind.
// TRACK = 1 (certified reuse): the UNGATED exact neighbour is stored together with lb2 = a lower bound on the
// distance from the transformed point to every other target point, for k_certify of the next pass.  A point
// that moved less than mu since the previous pass searches up to mu beyond its match to make that bound useful;
// one that moved more searches exactly like the plain kernel (its bound then carries no margin).
// RB (round 6): 1 = the rings of a plain point search prune by the target's row-block boxes (nn_rings_box); 2 = a target with
// heavy cells: every range longer than PCR_LB_MIN records is scanned through its leaf / group boxes (nn_scan_range_lb)
template <int VOXEL, int HALO, int TRACK, int RB = 0>
__device__ __forceinline__ void nn_point(const LinArgs &a, const Geom<float> &gf, const PoseK &P, const PoseQ &Q, int64_t i) {
    const float x = a.sx[i], y = a.sy[i], z = a.sz[i];
    float tx, ty, tz;
    xform(P, x, y, z, tx, ty, tz);
    float mu = 0.f;
    if (TRACK) {
        float ux, uy, uz;
        xform(Q, x, y, z, ux, uy, uz);
        const float m = __builtin_sqrtf(dist2_f32(tx - ux, ty - uy, tz - uz));
        mu = m < a.mu_f ? a.mu_f : 0.f;                    // (a NaN motion compares false: no margin)
    }
    uint32_t bj = PCR_NONE, bo = PCR_NONE;
    // a wave none of whose points moved little enough to be worth a margin runs the PLAIN search (second-best
    // tracking costs 20-37 % of a far-pose search): then every other point is no closer than the match, lb2 = d1
    const bool track = TRACK && __any(mu > 0.f);
    if (!VOXEL) {
        float best = a.bound2_f, lb2q;
        if (track) {
            NNTrack<float> tk;
            nn_track_init<float>(tk, a.bound2_f, mu);
            nn_search<float, PtF, false, false, HALO != 0, true>(gf, a.pts, a.cell_start, tx, ty, tz, a.bound2_f, best, bj, bo, nullptr, &tk);
            lb2q = fminf(tk.second, tk.pmin);
        } else {
#ifdef PCR_EXP_SEED
            // developer TIMING experiment (exact results): what the per-lane search costs when it starts from a near-exact
            // upper bound -- the match the previous pass left in nn_j (tools/pose0_passes.py repeats ONE pose, so that is the
            // true neighbour): the potential of any scheme that proposes a near neighbour before the exact search
            const uint32_t pj = a.nn_j[i];
            if (pj != PCR_NONE) nn_test<float, PtF, 0>(a.pts[pj], pj, tx, ty, tz, best, bj, bo);
            nn_search<float, PtF, false, true, HALO != 0, false>(gf, a.pts, a.cell_start, tx, ty, tz, a.bound2_f, best, bj, bo);
#else
            nn_search<float, PtF, false, false, HALO != 0, false, false, PCR_NN_BATCH, RB == 1, RB == 2>(gf, a.pts, a.cell_start, tx, ty, tz, a.bound2_f, best, bj, bo);
#endif
            lb2q = best;
        }
        if (TRACK) {
            a.nn_j[i] = bo != PCR_NONE ? bj : PCR_NONE;
            a.lb2[i] = __builtin_sqrtf(lb2q) * 0.99999f;
        } else {
            const bool ok = bo != PCR_NONE && __builtin_sqrtf(best) < a.md_f;
            nnj_store(a.nn_j + i, ok ? bj : PCR_NONE);
        }
    } else {
        double best = a.bound2_d, lb2q;
        if (track) {
            NNTrack<double> tk;
            nn_track_init<double>(tk, a.bound2_d, (double)mu);
            nn_search<double, PtD, false, false, false, true, VOXEL == 2>(a.gd, a.means, a.cell_start, (double)tx, (double)ty, (double)tz,
                                                             a.bound2_d, best, bj, bo, nullptr, &tk);
            lb2q = fmin(tk.second, tk.pmin);
        } else {
            nn_search<double, PtD, false, false, false, false, VOXEL == 2>(a.gd, a.means, a.cell_start, (double)tx, (double)ty, (double)tz,
                                                              a.bound2_d, best, bj, bo);
            lb2q = best;
        }
        if (TRACK) {
            a.nn_j[i] = bo != PCR_NONE ? bj : PCR_NONE;
            a.lb2[i] = (float)(__builtin_sqrt(lb2q) * 0.99999);
        } else {
            const bool ok = bo != PCR_NONE && __builtin_sqrt(best) < a.md_d;
            nnj_store(a.nn_j + i, ok ? bj : PCR_NONE);
        }
    }
}

// Plain centroid search, float32 FILTER + float64 check (round 3).  The float64 ring search costs 1.6x the float32
// one over the same centroids (32-byte records, half-rate arithmetic, no packed key).  So: a float32 TRACKING search over
// the centroids rounded to float32 (margin mu_ff, a fraction of a millimetre) returns a winner and a lower bound lbq on
// the float32-space distance of every other rounded centroid; rounding moved no centroid by more than `band`, so every
// other TRUE centroid is farther than lbq - band.  The winner's float64 distance d is computed exactly as the float64
// search computes it (nn_test<double>); if (lbq - band)^2 > d the float64 search would have returned this winner --
// strictly closer than everything else, so the tie rule is not involved.  "Nothing within the inflated bound" is
// certified the same way.  A lane that cannot certify (two centroids within ~0.1 mm of the same distance, duplicates)
// is left to k_nn_fix, the launch behind this one: nn_j = PCR_PENDING, and the pass' stamp goes into a.pending so that
// k_nn_fix returns at once when no lane asked (it runs the float64 search for the pending points: identical results by
// construction, `test_centroid_filter_is_exact`).  Inlining that search here instead cost 96 VGPRs + 140 spilled SGPRs.
#define PCR_PENDING_BIT 0x80000000u       // nn_j of a point the filter could not certify: this bit + the filter's nominee
__device__ __forceinline__ bool nn_is_pending(uint32_t j) { return (j & PCR_PENDING_BIT) != 0 && j != PCR_NONE; }
// SETTLE (round 4): the tracking search also carries the RUNNER-UP's index and a bound on everybody but the first two
// (NNTrack::third, nn_test_f32<2>).  A lane whose winner cannot be certified alone -- two rounded centroids within the
// margin of the same distance -- computes the runner-up's float64 distance too, takes the nearer of the two by the float64
// search's own rule (distance, then the smaller ORIGINAL index), and certifies THAT against the third bound.  Only a
// third candidate inside the margin is left pending.
// (the search + check for one transformed point: returns true when the answer is CERTIFIED -- w = cell-sorted index of the
// nearest centroid or PCR_NONE when nothing lies within the search bound, d = its squared float64 distance)
// Q6 (round 5; quirk Q6, plane_icp.py:20-22: a float64 PlaneICP target is SEARCHED in float64): the same filter with the
// target's OWN float32 index in front and the float64 coordinates of the same points, kept in the index's cell-sorted order
// (a.means = pcr_target::pts64), behind it -- the winner is addressed by its cell-sorted index fj, which is also what the
// reduce kernel gathers the float32 record with; its original index (the tie rule) sits in pts64[fj].w.
template <int HALO, int SETTLE, int B, int Q6>
__device__ __forceinline__ bool nn_filter_core(const LinArgs &a, float tx, float ty, float tz, uint32_t &w, double &d) {
    static_assert(!(Q6 && SETTLE), "two-way settling tracks original indices only");
    uint32_t fj = PCR_NONE, fo = PCR_NONE;
    float best = a.bound2_ff;
    NNTrack<float> tk;
    nn_track_init<float>(tk, a.bound2_ff, a.mu_ff);
    nn_search<float, PtF, false, false, HALO != 0, SETTLE ? 2 : 1, false, B>(a.gf, a.pts, a.cs_f, tx, ty, tz, a.bound2_ff, best, fj, fo, nullptr, &tk);
    w = PCR_NONE; d = 0.0;         // nothing within bound + band among the rounded centroids: nothing within the gate
    if (fo == PCR_NONE) return true;
    if (Q6) fo = fj;
    const PtD m = a.means[fo];
    const double dx = (double)tx - m.x, dy = (double)ty - m.y, dz = (double)tz - m.z;
    d = (dx * dx + dy * dy) + dz * dz;
    const float lbq = __builtin_sqrtf(fminf(tk.second, tk.pmin)) * 0.99999f - a.band_f;
    bool cert = lbq > 0.f && (double)lbq * (double)lbq > d * 1.000001;
    w = fo;
    if (SETTLE && !cert && tk.sec_o != PCR_NONE) {
        const PtD m2 = a.means[tk.sec_o];
        const double ex = (double)tx - m2.x, ey = (double)ty - m2.y, ez = (double)tz - m2.z;
        const double d2 = (ex * ex + ey * ey) + ez * ez;
        const bool take2 = (d2 < d) | ((d2 == d) & (pt_orig(m2) < pt_orig(m)));     // nn_test<double>'s rule
        w = take2 ? tk.sec_o : fo;
        d = take2 ? d2 : d;
        const float lb3 = __builtin_sqrtf(fminf(tk.third, tk.pmin)) * 0.99999f - a.band_f;
        cert = lb3 > 0.f && (double)lb3 * (double)lb3 > d * 1.000001;
    }
    return cert;
}

template <int HALO, int SETTLE, int Q6 = 0>
__device__ __forceinline__ void nn_point_filter(const LinArgs &a, const PoseK &P, int64_t i) {
    const float x = a.sx[i], y = a.sy[i], z = a.sz[i];
    float tx, ty, tz;
    xform(P, x, y, z, tx, ty, tz);
    uint32_t w; double d;
    const bool cert = nn_filter_core<HALO, SETTLE, PCR_NN_BATCH, Q6>(a, tx, ty, tz, w, d);
    uint32_t out = (w != PCR_NONE && __builtin_sqrt(d) < a.md_d) ? w : PCR_NONE;
#ifdef PCR_DEV
    if (!cert && !(a.flags & (2u << 28))) {          // (PCR_FIX_DEBUG=2, timing only: the nominee is taken unchecked)
#else
    if (!cert) {
#endif
        out = PCR_PENDING_BIT | w;                   // (cert is only ever false with a nominee: w != PCR_NONE)
        atomicMax(a.pending, a.stamp);
    }
    nnj_store(a.nn_j + i, out);
}

// The float64 answer for a point nn_point_filter could not certify.  The filter left its NOMINEE behind (nn_j =
// PCR_PENDING_BIT | index): the exact nearest centroid is the lexicographic minimum of (float64 distance, original index)
// over the centroids inside the ball through the nominee, which usually touches a handful of cells.  A from-scratch ring
// search bounded by the gate is a chain of ~50 dependent round trips (two per row of cells: entry, then records) --
// 50-65 us with the rest of the chip idle, whatever the number of pending points; here the ball's cell box is walked slab
// by slab, the entries of five rows requested together and the rows scanned with the nominee as the running best.
// Measured inside k_reduce_finalize<.., FIX> (developer build, PCR_FIX_DEBUG: stream only / + walk of nn_j / + searches):
// vplane_10m 72 / 83 / 125 us with the ring search for boxes wider than 3 x 3 rows, ndt_10m 100 / 111 / 118 us.
// (the box search itself: nearest record of `means` -- (float64 distance, original index) -- inside the ball through `nominee`)
__device__ __forceinline__ void nn_box_f64(const Geom<double> &g, const PtD *__restrict__ means, const uint32_t *__restrict__ cell_start,
                                           double qx, double qy, double qz, uint32_t nominee, double &bd, uint32_t &bj, uint32_t &bo) {
    const PtD mA = means[nominee];
    bj = nominee; bo = pt_orig(mA);
    {
        const double dx = qx - mA.x, dy = qy - mA.y, dz = qz - mA.z;
        bd = (dx * dx + dy * dy) + dz * dz;                           // nn_test<double>'s expression
    }
    const double r = __builtin_sqrt(bd) * 1.0000001 + g.slack;
    const double lim = 1.0e9;
    auto cell = [&](double v, double o, int n) {
        const double c = fmin(fmax((v - o) * g.inv_h, -lim), lim);
        return min(max((int)floor(c), 0), n - 1);
    };
    const int xl = cell(qx - r, g.ox, g.nx), xh = cell(qx + r, g.ox, g.nx);
    const int yl = cell(qy - r, g.oy, g.ny), yh = cell(qy + r, g.oy, g.ny);
    const int zl = cell(qz - r, g.oz, g.nz), zh = cell(qz + r, g.oz, g.nz);
    // slabs of the box in z, rows five at a time: their ten entries are requested together (one round trip), then the rows are
    // scanned with the running best; a slab or row that lies beyond the best is skipped (the ball only shrinks)
    const uint32_t unx = (uint32_t)g.nx, plane = (uint32_t)g.ny * unx;
    for (int z = zl; z <= zh; ++z) {
        const double zlo_ = g.oz + (double)z * g.h;
        const double dzm = fmax(fmax(zlo_ - qz, qz - (zlo_ + g.h)) - g.slack, 0.0);
        if (dzm * dzm > bd) continue;
        for (int y0 = yl; y0 <= yh; y0 += 5) {
            uint32_t s_[5], e_[5];
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const int y = y0 + k;
                const double ylo_ = g.oy + (double)y * g.h;
                const double dym = fmax(fmax(ylo_ - qy, qy - (ylo_ + g.h)) - g.slack, 0.0);
                const bool live = y <= yh && dzm * dzm + dym * dym <= bd;
                const uint32_t row = (uint32_t)z * plane + (uint32_t)(live ? y : yl) * unx;
                const uint32_t s0 = cell_start[row + (uint32_t)xl] & g.cs_mask, e0 = cell_start[row + (uint32_t)xh + 1u] & g.cs_mask;
                s_[k] = s0; e_[k] = live ? e0 : s0;
            }
#pragma unroll
            for (int k = 0; k < 5; ++k) nn_scan_range<double, PtD, 0>(means, s_[k], e_[k], qx, qy, qz, bd, bj, bo);
        }
    }
}

__device__ __forceinline__ void nn_point_fix(const LinArgs &a, const PoseK &P, int64_t i, uint32_t nominee) {
    float tx, ty, tz;
    xform(P, a.sx[i], a.sy[i], a.sz[i], tx, ty, tz);
    double bd;
    uint32_t bj, bo;
    nn_box_f64(a.gd, a.means, a.cell_start, (double)tx, (double)ty, (double)tz, nominee, bd, bj, bo);
    a.nn_j[i] = __builtin_sqrt(bd) < a.md_d ? bj : PCR_NONE;
}

// ---- fold the per-block partials in a fixed order and emit the 29-vector ---------------------
struct FinArgs {
    const double *partials;
    uint32_t *tile_ctr;        // the PCR_TILE_CTRS tile counters of k_nn_scan (re-armed by the fold)
    uint32_t *tickets;         // 64 B apart: [0..7] group tickets, [8] leader tickets
    int nblocks;
    int kind;
    double R[9];               // rotation of the pose when it came by value (pose == NULL)
    double *out;               // 32 doubles in HBM
    double *host_out;          // optional: the same 29 values straight into pinned host memory ...
    volatile uint32_t *host_flag;   // ... followed by this sequence number (host spins on it)
    uint32_t seq;
    // certified reuse: what the search of this pass did (PCR_NN_*, -1 = read it from the pose) and the per-block
    // counts of k_certify; reported in out[29] (points searched by a LIST pass) and out[30] (mode)
    int nn_mode;
    const uint32_t *ucnt;
    int n_ucnt;
    // bounding box of the scan and the displacement below which the next search deals its tiles block-locally
    float bb_c[3], bb_e[3];
    double local_len;
    double deep_len;           // ... and the displacement from which on it reads the deeper set of extended lists
    // device-resident Gauss-Newton loop (pcr_align; registration.py:89-111 behind the boundary)
    PoseDev *pose;             // NULL: plain pass
    int max_iter;
    double tol;
    double *trace;             // [max_iter][45]: pose before the step (16) + the 29 sums
    double *host_T;            // pinned: the pose after the step ...
    volatile unsigned long long *host_state;   // ... then (done << 32 | passes completed), one 8-byte store
};

// tot[0..31] (shared memory, complete before the call) -> the 29-vector in HBM and, optionally, in
// pinned host memory followed by the sequence number; also re-arms the tile counters.
__device__ __forceinline__ void finalize_emit(const FinArgs &f, const double *tot) {
    if (threadIdx.x < PCR_TILE_CTRS) f.tile_ctr[threadIdx.x * PCR_TILE_STRIDE] = 0;     // ready for the next k_nn_scan
    // points k_certify left to the search (LIST passes)
    __shared__ uint32_t listed;
    const int mode = f.nn_mode;
    if (mode == PCR_NN_LIST) {
        if (threadIdx.x == 0) listed = 0;
        __syncthreads();
        uint32_t v = 0;
        for (int b = threadIdx.x; b < f.n_ucnt; b += blockDim.x) v += f.ucnt[b];
        if (v) atomicAdd(&listed, v);
        __syncthreads();
    }
    // the 29-vector is assembled in LDS by one thread, then stored by 32: to HBM and, in parallel, to pinned host
    // memory (round 2 had thread 0 store 29 values to HBM and READ THEM BACK for 31 stores to the host, one after the other)
    __shared__ double o_sh[32];
    if (threadIdx.x == 0) {
        if (f.kind != PCR_ICP) {
            for (int i = 0; i < 29; ++i) o_sh[i] = tot[i];
        } else {
            // H_ll = M I (icp.py:43); H_lr = -R skew(sum p) (icp.py:44); H_rr from the second
            // moments (math_tools.py:44-58)
            const double *R = f.pose ? f.pose->R : f.R;
            const double cnt = tot[0], sx = tot[1], sy = tot[2], sz = tot[3];
            const double S[9] = {0, -sz, sy, sz, 0, -sx, -sy, sx, 0};
            double H[6][6];
            for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) H[i][j] = 0.0;
            H[0][0] = H[1][1] = H[2][2] = cnt;
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) {
                    double v = 0.0;
                    for (int k = 0; k < 3; ++k) v += R[3 * i + k] * S[3 * k + j];
                    H[i][3 + j] = -v;
                }
            const double xx = tot[4], xy = tot[5], xz = tot[6], yy = tot[7], yz = tot[8], zz = tot[9];
            H[3][3] = yy + zz; H[3][4] = -xy; H[3][5] = -xz;
            H[4][4] = xx + zz; H[4][5] = -yz; H[5][5] = xx + yy;
            int p = 0;
            for (int i = 0; i < 6; ++i) for (int j = i; j < 6; ++j) o_sh[p++] = H[i][j];
            for (int i = 0; i < 3; ++i) { o_sh[21 + i] = tot[10 + i]; o_sh[24 + i] = tot[13 + i]; }
            o_sh[27] = tot[16]; o_sh[28] = cnt;
        }
        o_sh[29] = mode == PCR_NN_LIST ? (double)listed : 0.0; o_sh[30] = (double)mode; o_sh[31] = 0;
    }
    __syncthreads();
    if (threadIdx.x < 32) {
        const double v = o_sh[threadIdx.x];
        f.out[threadIdx.x] = v;
        if (f.host_out) {
            f.host_out[threadIdx.x] = v;          // (the 32nd double of the pinned block is unused padding: the flag lives at [32])
            __threadfence_system();
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && f.host_out) *f.host_flag = f.seq;
}

// The fold of the per-block partial sums INSIDE the producing kernel (no separate k_finalize launch: ~10 us
// and a launch gap per pass).  Any grid that is a multiple of 8 blocks; two levels of tickets.  Blocks g,
// g+8, g+16, ... form group g (the blocks the dispatcher places on XCD g, so a group's traffic stays in one
// L2 -- a locality assumption only, every cross-block access is coherent at agent scope).  The block that
// takes a group's last ticket folds the group's partials into row nblocks+g; the group leader that takes the
// last of the 8 second-level tickets folds those rows and emits.  8 x (nblocks/8) + 8 serialised atomics
// instead of nblocks.
// (returns true in the ONE block that folded the last contribution and emitted the result)
template <bool SPLIT = false>
__device__ __forceinline__ bool ticket_fold_emit(double *acc, const LinArgs &a, const FinArgs &f) {
    block_store_partials<true, SPLIT>(acc, a.partials);

    __shared__ int role;
    __shared__ double part[8][33];
    __shared__ double tot[32];
    const int ng = 8;          // (a single group for small grids was measured: no gain, 15.9 vs 15.7 us at 100 k points)
    const int g = (int)(blockIdx.x & 7), per = f.nblocks / ng;
    uint32_t *ctr1 = &f.tickets[g * PCR_TICKET_STRIDE], *ctr2 = &f.tickets[8 * PCR_TICKET_STRIDE];
    double *rows = const_cast<double *>(f.partials);
    // Hand-off protocol (MI355X guide, "sc1 payload -> drained vmcnt -> sc1 flag"): the 32 partial sums
    // were stored write-through at agent scope by lanes 0..31 of THIS wave; the explicit s_waitcnt below
    // (inline asm: the compiler cannot drop or move it) makes the wave wait until those stores have
    // been acknowledged by memory before the ticket atomic is issued, so a block on another XCD that
    // observes the ticket also observes the rows.  The folding block reads the rows with agent-scope
    // (sc1, L1-bypassing) loads issued after its own ticket returned.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0) {
        const uint32_t t = __hip_atomic_fetch_add(ctr1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        role = t == (uint32_t)(per - 1);
    }
    __syncthreads();
    if (!role) return false;

    // ---- group leader: rows g + ng i, i = 0 .. per-1, in a fixed order; 16 loads in flight per thread
    const int c = threadIdx.x & 31, r0 = threadIdx.x >> 5;
    double s16[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) s16[u] = 0.0;
    for (int i0 = 0; i0 < per; i0 += 128) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int i = i0 + r0 + 8 * u;
            double v = 0.0;
            if (i < per) v = __hip_atomic_load(&rows[(size_t)(g + ng * i) * 32 + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s16[u] += v;
        }
    }
    part[r0][c] = (((s16[0] + s16[1]) + (s16[2] + s16[3])) + ((s16[4] + s16[5]) + (s16[6] + s16[7]))) +
                  (((s16[8] + s16[9]) + (s16[10] + s16[11])) + ((s16[12] + s16[13]) + (s16[14] + s16[15])));
    __syncthreads();
    if (threadIdx.x < 32) {
        const int k = threadIdx.x;
        const double t = ((part[0][k] + part[1][k]) + (part[2][k] + part[3][k])) + ((part[4][k] + part[5][k]) + (part[6][k] + part[7][k]));
        __hip_atomic_store(&rows[(size_t)(f.nblocks + g) * 32 + k], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // group row stored (lanes 0..31 of wave 0)
    if (threadIdx.x == 0) {
        __hip_atomic_store(ctr1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // re-armed for the next pass
        const uint32_t t2 = __hip_atomic_fetch_add(ctr2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        role = t2 == 7u;
    }
    __syncthreads();
    if (!role) return false;

    // ---- the last group leader: the 8 group rows, in order
    if (threadIdx.x == 0) __hip_atomic_store(ctr2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x < 32) {
        double v[8];
#pragma unroll
        for (int gg = 0; gg < 8; ++gg)
            v[gg] = __hip_atomic_load(&rows[(size_t)(f.nblocks + gg) * 32 + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tot[threadIdx.x] = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
    __syncthreads();
    finalize_emit(f, tot);
    return true;
}

// ---- developer / A-B kernels (kernels_dev.hip): unfused folds, the wave-cooperative search, work counters ----
void pcr_dev_launch_linearize(int kind, bool halo, dim3 grid, hipStream_t st, const LinArgs &a);
void pcr_dev_launch_reduce(int kind, dim3 grid, hipStream_t st, const LinArgs &a);
void pcr_dev_launch_finalize(hipStream_t st, const FinArgs &f);
void pcr_dev_launch_coop(dim3 grid, hipStream_t st, const LinArgs &a);
int pcr_dev_coop_blocks_per_cu();
int choose_blocks(const pcr_context *ctx, int64_t n);
