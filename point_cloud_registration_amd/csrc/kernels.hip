// Hot-path kernels: transform -> exact NN -> gate -> residual/Jacobian -> 6x6 normal equations -> (in
// pcr_align) the Gauss-Newton step itself.
//
// One calc_H_g_e2 of the reference (icp.py:24-57, plane_icp.py:30-69,
// voxelized_plane_icp.py:23-64, ndt.py:24-57) = k_nn_scan + k_reduce_finalize<KIND>  (shipped pipeline;
// k_linearize + k_finalize, k_reduce + k_finalize and k_nn_coop are kept selectable for A/B runs).
// One iteration of Registration.align (registration.py:89-111) = that + k_gn_update.
//
// Data layout in HBM
//   scan      SoA x[], y[], z[] float32, Morton-sorted once per align()  -> 3 coalesced dword
//             streams, 12 B/point, neighbouring lanes are neighbouring points in space;
//             nn_j u32[]: the matched cell-sorted index of every scan point (search -> reduce)
//   target    cell-sorted float4 {x, y, z, orig idx} for the search; cell-sorted 32-byte
//             {point, normal} records for the PlaneICP gather; cell_start u32[ncells+1]; optional
//             halo lists (nn_device.h)   (voxel targets: double4 means, double[3] normals,
//             double[6] inverse covariances; a few MB, L2-resident)
//   output    per-block partial sums [nblocks + 8][32] double, folded in fixed order inside
//             k_reduce_finalize (deterministic: no floating-point atomics anywhere)
//
// Roofline: HBM-bound gather/stream work, no dense contraction -> no MFMA.  Algorithmic bytes
// per scan point: ICP 24, PlaneICP 36, VPlaneICP 36, NDT 48.
//
// Launch: 256-thread blocks (4 waves of 64).  The sorted scan is split into contiguous spans per XCD
// (block b runs on XCD b % 8), so each XCD sweeps one region of space and its private 4 MiB L2 holds
// that region's target cells (see TileIter / nn_tile_loop).  Per-lane accumulators are float64 (H
// entries reach 1e11 at 1e8 points, float32 would lose the 1e-5 parity bar); the 32 sums are folded
// across the wave with a halving butterfly (32 shuffles instead of 32 x 6), then across waves through LDS.
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
    unsigned flags;
    int nblocks;
    double *partials;  // [nblocks + 8][32]
    // variant 1: correspondences through HBM
    uint32_t *nn_j;
    uint32_t *tile_ctr;   // tile counters (64 B apart) of the NN kernels' dynamic hand-out
    int sched_local;      // 1: block-local hand-out (small scans), 0: global counters (see nn_tile_loop)
};

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
    // J = [I, -R skew(p)] (ndt.py:40); C symmetric inverse covariance
    const double C[3][3] = {{c6[0], c6[1], c6[2]}, {c6[1], c6[3], c6[4]}, {c6[2], c6[4], c6[5]}};
    double J[3][6];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double ri0 = a.R[3 * i], ri1 = a.R[3 * i + 1], ri2 = a.R[3 * i + 2];
        J[i][0] = i == 0; J[i][1] = i == 1; J[i][2] = i == 2;
        // -(R S) with S = [[0,-z,y],[z,0,-x],[-y,x,0]]
        J[i][3] = -(ri1 * z - ri2 * y);
        J[i][4] = -(-ri0 * z + ri2 * x);
        J[i][5] = -(ri0 * y - ri1 * x);
    }
    double CJ[3][6], Cd[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        Cd[i] = C[i][0] * d0 + C[i][1] * d1 + C[i][2] * d2;
#pragma unroll
        for (int j = 0; j < 6; ++j) CJ[i][j] = C[i][0] * J[0][j] + C[i][1] * J[1][j] + C[i][2] * J[2][j];
    }
    int p = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = i; j < 6; ++j) { acc[p] += J[0][i] * CJ[0][j] + J[1][i] * CJ[1][j] + J[2][i] * CJ[2][j]; ++p; }
#pragma unroll
    for (int i = 0; i < 6; ++i) acc[21 + i] += J[0][i] * Cd[0] + J[1][i] * Cd[1] + J[2][i] * Cd[2];
    acc[27] += d0 * Cd[0] + d1 * Cd[1] + d2 * Cd[2];
    acc[28] += 1.0;
}

// the gate of the reference (icp.py:34, plane_icp.py:41, voxelized_plane_icp.py:38, ndt.py:33: dist < max_dist, strict), on
// the distance exactly as the search computes it (nn_test): the reduce kernels apply it themselves, so that the
// search may leave UNGATED matches behind for the next pass (certified reuse)
__device__ __forceinline__ bool gate_f32(const LinArgs &a, float dx, float dy, float dz) {
    return __builtin_sqrtf(dist2_f32(dx, dy, dz)) < a.md_f;
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
template <int KIND>
__device__ __forceinline__ void reduce_stream(double *acc, const LinArgs &a, const PoseK &P, int64_t base, int64_t end,
                                              int64_t stride) {
    if (KIND == PCR_ICP || KIND == PCR_PLANE) {
        for (int64_t i = base; i < end; i += 2 * stride) {
            const int64_t i1 = i + stride;
            const bool two = i1 < end;
            const uint32_t j0 = a.nn_j[i];
            const uint32_t j1 = two ? a.nn_j[i1] : PCR_NONE;
            const bool ok0 = j0 != PCR_NONE, ok1 = j1 != PCR_NONE;
            float4 q0 = make_float4(0, 0, 0, 0), n0 = q0, q1 = q0, n1 = q0;
            if (KIND == PCR_PLANE) {
                if (ok0) { const float4 *r = reinterpret_cast<const float4 *>(a.pn + j0); q0 = r[0]; n0 = r[1]; }
                if (ok1) { const float4 *r = reinterpret_cast<const float4 *>(a.pn + j1); q1 = r[0]; n1 = r[1]; }
            } else {
                if (ok0) q0 = a.pts[j0];
                if (ok1) q1 = a.pts[j1];
            }
            if (ok0) {
                const float x = a.sx[i], y = a.sy[i], z = a.sz[i];
                float tx, ty, tz;
                xform(P, x, y, z, tx, ty, tz);
                const float dx = tx - q0.x, dy = ty - q0.y, dz = tz - q0.z;
                if (gate_f32(a, dx, dy, dz)) {
                    if (KIND == PCR_PLANE) acc_plane(acc, P, x, y, z, n0.x, n0.y, n0.z, (double)dx, (double)dy, (double)dz);
                    else acc_icp(acc, P, a.flags, x, y, z, (double)dx, (double)dy, (double)dz);
                }
            }
            if (ok1) {
                const float x = a.sx[i1], y = a.sy[i1], z = a.sz[i1];
                float tx, ty, tz;
                xform(P, x, y, z, tx, ty, tz);
                const float dx = tx - q1.x, dy = ty - q1.y, dz = tz - q1.z;
                if (gate_f32(a, dx, dy, dz)) {
                    if (KIND == PCR_PLANE) acc_plane(acc, P, x, y, z, n1.x, n1.y, n1.z, (double)dx, (double)dy, (double)dz);
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
__device__ __forceinline__ void wave_fold32(double *acc, int lane) {
    fold_step<16, 32>(acc, lane);
    fold_step<8, 16>(acc, lane);
    fold_step<4, 8>(acc, lane);
    fold_step<2, 4>(acc, lane);
    fold_step<1, 2>(acc, lane);
    acc[0] += shfl_xor_f64(acc[0], 1);
}

// COHERENT: the store is written through to memory at agent scope, so that a block on ANOTHER XCD
// (each XCD has a private, mutually non-coherent L2) can read it inside the same kernel.
template <bool COHERENT = false>
__device__ __forceinline__ void block_store_partials(double *acc, double *__restrict__ partials) {
    __shared__ double wsum[4][32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    wave_fold32(acc, lane);
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
template <int KIND, int HALO>
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
            nn_search<float, PtF, false, false, HALO != 0>(a.gf, a.pts, a.cell_start, tx, ty, tz, a.bound2_f, best, bj, bo);
            ok = bj != PCR_NONE && __builtin_sqrtf(best) < a.md_f;                 // icp.py:34 strict gate
        } else {
            double best;
            nn_search<double, PtD>(a.gd, a.means, a.cell_start, (double)tx, (double)ty, (double)tz, a.bound2_d, best, bj, bo);
            ok = bj != PCR_NONE && __builtin_sqrt(best) < a.md_d;                 // voxelized_plane_icp.py:38
        }
        if (ok) accumulate<KIND, false>(acc, a, P, bj, x, y, z, tx, ty, tz);
    }
}

template <int KIND, int HALO>
__global__ void __launch_bounds__(256) k_linearize(const LinArgs a) {
    PoseK P;
    if (!load_pose<true>(a, P)) return;
    double acc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = 0.0;
    linearize_body<KIND, HALO>(a, P, acc);
    block_store_partials(acc, a.partials);
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
#define PCR_TILE_STRIDE 16
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
}

// One query: scan point i, on its own lane (gathers): the general search.  HALO: the target has the extended
// per-cell lists and ring 0 reads those (nn_ring0).
// TRACK = 0: the match is gated here (PCR_NONE = no correspondence) -- nothing else is left behind.
// TRACK = 1 (certified reuse): the UNGATED exact neighbour is stored together with lb2 = a lower bound on the
// distance from the transformed point to every other target point, for k_certify of the next pass.  A point
// that moved less than mu since the previous pass searches up to mu beyond its match to make that bound useful;
// one that moved more searches exactly like the plain kernel (its bound then carries no margin).
template <int VOXEL, int HALO, int TRACK>
__device__ __forceinline__ void nn_point(const LinArgs &a, const PoseK &P, const PoseQ &Q, int64_t i) {
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
            nn_search<float, PtF, false, false, HALO != 0, true>(a.gf, a.pts, a.cell_start, tx, ty, tz, a.bound2_f, best, bj, bo, nullptr, &tk);
            lb2q = fminf(tk.second, tk.pmin);
        } else {
            nn_search<float, PtF, false, false, HALO != 0, false>(a.gf, a.pts, a.cell_start, tx, ty, tz, a.bound2_f, best, bj, bo);
            lb2q = best;
        }
        if (TRACK) {
            a.nn_j[i] = bo != PCR_NONE ? bj : PCR_NONE;
            a.lb2[i] = __builtin_sqrtf(lb2q) * 0.99999f;
        } else {
            const bool ok = bo != PCR_NONE && __builtin_sqrtf(best) < a.md_f;
            a.nn_j[i] = ok ? bj : PCR_NONE;
        }
    } else {
        double best = a.bound2_d, lb2q;
        if (track) {
            NNTrack<double> tk;
            nn_track_init<double>(tk, a.bound2_d, (double)mu);
            nn_search<double, PtD, false, false, false, true>(a.gd, a.means, a.cell_start, (double)tx, (double)ty, (double)tz,
                                                             a.bound2_d, best, bj, bo, nullptr, &tk);
            lb2q = fmin(tk.second, tk.pmin);
        } else {
            nn_search<double, PtD, false, false, false, false>(a.gd, a.means, a.cell_start, (double)tx, (double)ty, (double)tz,
                                                              a.bound2_d, best, bj, bo);
            lb2q = best;
        }
        if (TRACK) {
            a.nn_j[i] = bo != PCR_NONE ? bj : PCR_NONE;
            a.lb2[i] = (float)(__builtin_sqrt(lb2q) * 0.99999);
        } else {
            const bool ok = bo != PCR_NONE && __builtin_sqrt(best) < a.md_d;
            a.nn_j[i] = ok ? bj : PCR_NONE;
        }
    }
}

// ---- certified reuse of the previous pass' matches ------------------------------------------------
// Registration.align (registration.py:89-111) repeats the full search every iteration although the converged
// tail moves the scan by millimetres.  Between two passes over the same scan and target:
//   * the previous pass left, per scan point, its exact nearest neighbour x1 (nn_j) and lb2 <= |q - y| for every
//     other target point y, q being the point under the previous pose (a tracking search: nn_point<TRACK>);
//   * under the new pose the point sits at q', m = |q' - q| away, so |q' - y| >= lb2 - m for every y != x1
//     (triangle inequality);  if |q' - x1| < lb2 - m, x1 is still the strict, unique nearest neighbour --
//     exactly what a fresh search would return -- and lb2 - m is the new bound;
//   * a point that had nothing inside the search bound keeps that state while lb2 - m stays above the gate.
// k_certify evaluates this for every point (one gather of the old match, no search), writes the new bounds and
// a bit mask of the points it could NOT certify; k_nn_scan<LIST> then searches only those, compacted so that the
// lanes of a wave stay dense; the reduce kernel is the one of every other pass, so the sums are bit-identical to
// a pass that searched everything.  Everything is float32 arithmetic on the float32 positions the search itself
// uses; the relative slacks (1e-5) cover the rounding of the distances (~1e-7) many times over.
template <int VOXEL>
__global__ void __launch_bounds__(256) k_certify(const LinArgs a) {
    PoseK P;
    PoseQ Q;
    if (!load_pose<false>(a, P)) return;
    load_prev(a, Q);
    const TileIter it(a);
    const int lane = threadIdx.x & 63;
    uint32_t marked = 0;
    for (int64_t i0 = it.base - threadIdx.x; i0 < it.end; i0 += it.stride) {
        const int64_t i = i0 + threadIdx.x;
        const bool live = i < it.end;
        bool cert = false;
        if (live) {
            const uint32_t j = a.nn_j[i];
            const float lb = a.lb2[i];
            const float x = a.sx[i], y = a.sy[i], z = a.sz[i];
            float tx, ty, tz, ux, uy, uz;
            xform(P, x, y, z, tx, ty, tz);
            xform(Q, x, y, z, ux, uy, uz);
            const float m = __builtin_sqrtf(dist2_f32(tx - ux, ty - uy, tz - uz)) * 1.00001f;
            const float lbn = (lb - m) * 0.999999f;
            if (j != PCR_NONE) {
                float d1;
                if (!VOXEL) {
                    const PtF q = a.pts[j];
                    d1 = __builtin_sqrtf(dist2_f32(tx - q.x, ty - q.y, tz - q.z));
                } else {
                    const PtD q = a.means[j];
                    const double dx = (double)tx - q.x, dy = (double)ty - q.y, dz = (double)tz - q.z;
                    d1 = (float)__builtin_sqrt((dx * dx + dy * dy) + dz * dz);
                }
                cert = d1 * 1.00002f < lbn;
            } else {
                cert = lbn > a.md_f * 1.00001f;            // still nothing inside the gate
            }
            if (cert) a.lb2[i] = lbn;
        }
        const unsigned long long mask = __ballot(live && !cert);
        const int64_t w0 = i0 + (threadIdx.x & ~63);          // first point of this wave's tile
        if (lane == 0 && w0 < it.end) a.umask[w0 >> 6] = mask;
        marked += (uint32_t)__popcll(mask);
    }
    __shared__ uint32_t blk_marked;
    if (threadIdx.x == 0) blk_marked = 0;
    __syncthreads();
    if (lane == 0) atomicAdd(&blk_marked, marked);
    __syncthreads();
    if (threadIdx.x == 0) a.ucnt[blockIdx.x] = blk_marked;
}

// A chunk of a LIST pass: PCR_LIST_CHUNK consecutive scan points = 16 mask words, handed to a BLOCK.  Wave 0 expands
// the set bits into a list in LDS (lane l takes 16 bits: word l >> 2, quarter l & 3); the block's four waves then
// search the listed points 64 at a time, round-robin -- the lanes stay dense however few points k_certify left
// over, and a chunk with everything marked still runs four rounds per wave, like a full search.
// (First version: one wave per chunk, 16 rounds in sequence -- 4.6x slower than the full search when nothing
// certified.)
#define PCR_LIST_CHUNK 1024
template <int VOXEL, int HALO>
__device__ __forceinline__ void nn_chunk_list(const LinArgs &a, const PoseK &P, const PoseQ &Q, uint16_t *lst, uint32_t *lst_n,
                                              int64_t first, int64_t end) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave == 0) {
        const int64_t w = (first >> 6) + (lane >> 2);
        const int64_t nwords = (a.n + 63) >> 6;
        unsigned long long word = 0;
        if (w < nwords && (w << 6) < end) word = a.umask[w];
        uint32_t bits = (uint32_t)(word >> (16 * (lane & 3))) & 0xffffu;
        const uint32_t cnt = (uint32_t)__popc(bits);
        uint32_t incl = cnt;                                        // inclusive prefix sum over the wave
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t v = (uint32_t)__shfl_up((int)incl, d, 64);
            if (lane >= d) incl += v;
        }
        if (lane == 63) *lst_n = incl;
        uint32_t pos = incl - cnt;
        while (bits) {
            const uint32_t b = (uint32_t)__builtin_ctz(bits);
            lst[pos++] = (uint16_t)((lane << 4) | b);
            bits &= bits - 1;
        }
    }
    __syncthreads();
    const uint32_t total = *lst_n;
    for (uint32_t e = threadIdx.x; e < total; e += 256) nn_point<VOXEL, HALO, 1>(a, P, Q, first + (int64_t)lst[e]);
    __syncthreads();                                                // (the list is rewritten by the next chunk)
}

// block-level hand-out of the chunks of a LIST pass: the XCD's span of the scan is dealt to the XCD's blocks through
// one counter per XCD sub-span (same counters and the same sub-spans as nn_tile_loop; a block asks once per 1024
// points, so the atomics do not matter here)
template <typename Body>
__device__ __forceinline__ void nn_chunk_loop(const LinArgs &a, Body &&body) {
    const int xcd = (int)(blockIdx.x & 7);
    const uint32_t xb = blockIdx.x >> 3;
    const int64_t gspan = (((a.n + PCR_TILE_CTRS - 1) / PCR_TILE_CTRS) + (PCR_LIST_CHUNK - 1)) & ~(int64_t)(PCR_LIST_CHUNK - 1);
    __shared__ uint32_t chunk_t;
    for (int r = 0; r < PCR_TILE_SUB; ++r) {
        const int c = xcd + 8 * (int)((xb + r) % PCR_TILE_SUB);
        const int64_t lo = gspan * c;
        const int64_t end = lo + gspan < a.n ? lo + gspan : a.n;
        for (;;) {
            if (threadIdx.x == 0) chunk_t = atomicAdd(&a.tile_ctr[c * PCR_TILE_STRIDE], 1u);
            __syncthreads();
            const int64_t first = lo + (int64_t)chunk_t * PCR_LIST_CHUNK;
            __syncthreads();
            if (first >= end) break;
            body(first, first + PCR_LIST_CHUNK < end ? first + PCR_LIST_CHUNK : end);
        }
    }
}

// MODE: PCR_NN_FULL / PCR_NN_TRACK / PCR_NN_LIST
// (5 waves per SIMD: the float64 centroid search sits at 101 VGPRs without the bound, which would cost it a
// fifth of its occupancy and 10 % of its speed; the float32 search needs < 80 either way)
template <int VOXEL, int HALO, int LOCAL, int MODE>
__global__ void __launch_bounds__(256, 5) k_nn_scan(const LinArgs a) {
    PoseK P;
    PoseQ Q;
    if (!load_pose<false>(a, P)) return;
    if (MODE != PCR_NN_FULL) load_prev(a, Q);
    if (MODE == PCR_NN_LIST) {
        __shared__ uint16_t lst[PCR_LIST_CHUNK];
        __shared__ uint32_t lst_n;
        nn_chunk_loop(a, [&](int64_t first, int64_t end) { nn_chunk_list<VOXEL, HALO>(a, P, Q, lst, &lst_n, first, end); });
    } else {
        nn_tile_loop<LOCAL, 64>(a, [&](int64_t first, int64_t end) {
            const int64_t i = first + (threadIdx.x & 63);
            if (i < end) nn_point<VOXEL, HALO, MODE == PCR_NN_TRACK>(a, P, Q, i);
        });
    }
}

// host-side choice of the instantiation
template <int VOXEL, int MODE>
static void launch_nn_scan_mode(bool halo, bool local, dim3 grid, hipStream_t st, const LinArgs &a) {
    const dim3 block(256);
#define PCR_NN_CASE(H, L) hipLaunchKernelGGL((k_nn_scan<VOXEL, (VOXEL ? 0 : H), L, MODE>), grid, block, 0, st, a)
    if (halo) { if (local) PCR_NN_CASE(1, 1); else PCR_NN_CASE(1, 0); }
    else { if (local) PCR_NN_CASE(0, 1); else PCR_NN_CASE(0, 0); }
#undef PCR_NN_CASE
}
template <int VOXEL>
static void launch_nn_scan(int mode, bool halo, bool local, dim3 grid, hipStream_t st, const LinArgs &a) {
    switch (mode) {
    case PCR_NN_FULL: launch_nn_scan_mode<VOXEL, PCR_NN_FULL>(halo, local, grid, st, a); break;
    case PCR_NN_TRACK: launch_nn_scan_mode<VOXEL, PCR_NN_TRACK>(halo, local, grid, st, a); break;
    default: launch_nn_scan_mode<VOXEL, PCR_NN_LIST>(halo, false, grid, st, a); break;
    }
}

// ---- wave-cooperative search (point targets) ---------------------------------------------------
// The 64 queries of a tile are Morton neighbours moved by ONE rigid transform, so their search balls
// overlap almost entirely.  Instead of 64 lanes gathering 64 different candidate lists (divergent
// loops, one cache line per lane and load), the wave walks the rows of cells of the box spanned by all
// its balls ONCE: every candidate is fetched with a wave-uniform address (one line, served to all
// lanes) and tested by all 64 lanes; a row is skipped when no lane's bound reaches it.  No divergence,
// no per-lane gathers.  The ball of a lane comes from an exact upper bound: its match of the previous
// pass (seed) or, without one, whatever a first round over the lanes' own cells found.  Exactness: a
// lane is certified when the ball of its final best lies inside a box whose needed rows were all
// walked; anything else (box too large, too many candidates, still uncertified) goes to the per-lane
// search, started from the best found so far.
#ifndef PCR_COOP_CAP
#define PCR_COOP_CAP 512        // staged points per wave (16 B each): 4 waves x 8 KB of LDS per block
#endif
#ifndef PCR_COOP_LDS
#define PCR_COOP_LDS 1          // 1: candidates staged in LDS (async global->LDS copies); 0: uniform global loads
#endif
#if PCR_COOP_LDS
#define PCR_COOP_MAX_CAND PCR_COOP_CAP
#else
#define PCR_COOP_MAX_CAND 1536
#endif
typedef __attribute__((address_space(1))) const void *gas_ptr;
typedef __attribute__((address_space(3))) void *las_ptr;
extern "C" __device__ int __ockl_wfred_min_i32(int);
extern "C" __device__ int __ockl_wfred_max_i32(int);
extern "C" __device__ unsigned __ockl_wfred_add_u32(unsigned);

struct BallBox { int x0, x1, y0, y1, z0, z1; };

// cells a ball of radius r around the query can reach, clamped to the grid (conservative: slack)
__device__ __forceinline__ BallBox ball_cells(const Geom<float> &g, float tx, float ty, float tz, float r) {
    BallBox b;
    const float fx = (float)(g.nx - 1), fy = (float)(g.ny - 1), fz = (float)(g.nz - 1);
    b.x0 = (int)fminf(fmaxf(floorf((tx - r - g.ox) * g.inv_h), 0.f), fx);
    b.x1 = (int)fminf(fmaxf(floorf((tx + r - g.ox) * g.inv_h), 0.f), fx);
    b.y0 = (int)fminf(fmaxf(floorf((ty - r - g.oy) * g.inv_h), 0.f), fy);
    b.y1 = (int)fminf(fmaxf(floorf((ty + r - g.oy) * g.inv_h), 0.f), fy);
    b.z0 = (int)fminf(fmaxf(floorf((tz - r - g.oz) * g.inv_h), 0.f), fz);
    b.z1 = (int)fminf(fmaxf(floorf((tz + r - g.oz) * g.inv_h), 0.f), fz);
    return b;
}

template <int SEED>
__device__ __forceinline__ void nn_tile_coop(const LinArgs &a, const PoseK &P, PtF *stage, int64_t first, int64_t end) {
    const Geom<float> &g = a.gf;
    const int lane = threadIdx.x & 63;
    const int64_t i = first + lane;
    const bool exists = i < end;
    float x = 0.f, y = 0.f, z = 0.f;
    uint32_t pj = PCR_NONE;
    if (exists) {
        x = a.sx[i]; y = a.sy[i]; z = a.sz[i];
        if (SEED) pj = a.nn_j[i];
    }
    float tx, ty, tz;
    xform(P, x, y, z, tx, ty, tz);
    // NaN / inf queries match nothing (their distance never passes the gate)
    const bool live = exists && fabsf(tx) <= 3.0e38f && fabsf(ty) <= 3.0e38f && fabsf(tz) <= 3.0e38f;
    float best = a.bound2_f;
    uint32_t bj = PCR_NONE, bo = PCR_NONE;
    if (SEED && live && pj != PCR_NONE) nn_test<float, PtF>(a.pts[pj], pj, tx, ty, tz, best, bj, bo);
    const NNCell<float> c = nn_cell<float>(g, tx, ty, tz, a.bound2_f);
    const float rmax = __builtin_sqrtf(a.bound2_f) * 1.000002f + g.slack;
    const uint32_t unx = (uint32_t)g.nx, uny = (uint32_t)g.ny;
    bool pending = live;
    for (int round = 0; round < 3; ++round) {
        if (!__any(pending)) break;
        // this round's box: union of the pending lanes' balls (round 0, nothing found yet: the own cell)
        const bool has = best < a.bound2_f;
        const float r = has ? RealTraits<float>::sqrt_fast(best) * 1.000002f + g.slack : (round == 0 ? 0.f : rmax);
        const BallBox b = ball_cells(g, tx, ty, tz, r);
        const int X0 = __ockl_wfred_min_i32(pending ? b.x0 : 0x7fffffff), X1 = __ockl_wfred_max_i32(pending ? b.x1 : -1);
        const int Y0 = __ockl_wfred_min_i32(pending ? b.y0 : 0x7fffffff), Y1 = __ockl_wfred_max_i32(pending ? b.y1 : -1);
        const int Z0 = __ockl_wfred_min_i32(pending ? b.z0 : 0x7fffffff), Z1 = __ockl_wfred_max_i32(pending ? b.z1 : -1);
        const int by = Y1 - Y0 + 1, bz = Z1 - Z0 + 1, rows = by * bz;
        bool coop = by > 0 && bz > 0 && rows <= 64;
        uint32_t rs = 0, re = 0;
        if (coop) {
            if (lane < rows) {                                   // lane r fetches the point range of row r
                const uint32_t ry = (uint32_t)(Y0 + lane % by), rz = (uint32_t)(Z0 + lane / by);
                const uint32_t rowb = (rz * uny + ry) * unx;
                rs = a.cell_start[rowb + (uint32_t)X0] & g.cs_mask;
                re = a.cell_start[rowb + (uint32_t)X1 + 1u] & g.cs_mask;
            }
            coop = __ockl_wfred_add_u32(re - rs) <= PCR_COOP_MAX_CAND;
        }
        if (!coop) break;                                        // the per-lane search takes over below
#if PCR_COOP_LDS
        // ---- stage the rows of the box in LDS: asynchronous global->LDS copies, all in flight at once,
        // ONE wait; the walk below then reads candidates as LDS broadcasts (~100 cycles instead of an L2
        // round trip per batch)
        {
            uint32_t off = 0;
            for (int r2 = 0; r2 < rows; ++r2) {
                const uint32_t s_ = (uint32_t)__builtin_amdgcn_readlane((int)rs, r2);
                const uint32_t len = (uint32_t)__builtin_amdgcn_readlane((int)re, r2) - s_;
                for (uint32_t o = 0; o < len; o += 64) {
                    if (o + (uint32_t)lane < len)
                        __builtin_amdgcn_global_load_lds((gas_ptr)(a.pts + s_ + o + lane), (las_ptr)(stage + off + o), 16, 0, 0);
                }
                off += len;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
#endif
        int rr = 0;
        uint32_t roff = 0;                                       // LDS position of the current row's first point
        for (int zz = Z0; zz <= Z1; ++zz) {
            const int dzc = zz - c.cz;
            float dzm = dzc == 0 ? 0.f : (dzc > 0 ? (float)dzc * g.h - c.fz : (float)(-dzc - 1) * g.h + c.fz);
            dzm = fmaxf(dzm - g.slack, 0.f);
            const float dz2 = dzm * dzm;
            const bool zneed = __any(pending && dz2 <= best);
            for (int yy = Y0; yy <= Y1; ++yy, ++rr) {
                const uint32_t s_ = (uint32_t)__builtin_amdgcn_readlane((int)rs, rr);
                const uint32_t e_ = (uint32_t)__builtin_amdgcn_readlane((int)re, rr);
                const uint32_t base = roff;
                roff += e_ - s_;
                if (s_ == e_ || !zneed) continue;
                const int dyc = yy - c.cy;
                float dym = dyc == 0 ? 0.f : (dyc > 0 ? (float)dyc * g.h - c.fy : (float)(-dyc - 1) * g.h + c.fy);
                dym = fmaxf(dym - g.slack, 0.f);
                const float dyz2 = dz2 + dym * dym;
                if (!__any(pending && dyz2 <= best)) continue;
#if PCR_COOP_LDS
                const PtF *q = stage + base;                     // wave-uniform LDS address: broadcast reads
                uint32_t j = s_;
                for (; j + 4 <= e_; j += 4, q += 4) {
                    const PtF p0 = q[0], p1 = q[1], p2 = q[2], p3 = q[3];
                    nn_test<float, PtF>(p0, j, tx, ty, tz, best, bj, bo);
                    nn_test<float, PtF>(p1, j + 1, tx, ty, tz, best, bj, bo);
                    nn_test<float, PtF>(p2, j + 2, tx, ty, tz, best, bj, bo);
                    nn_test<float, PtF>(p3, j + 3, tx, ty, tz, best, bj, bo);
                }
                for (; j < e_; ++j, ++q) nn_test<float, PtF>(q[0], j, tx, ty, tz, best, bj, bo);
#else
                (void)base;
                for (uint32_t j = s_; j < e_; j += 4) {          // wave-uniform addresses: one line for all lanes
                    const PtF *__restrict__ q = a.pts + j;
                    const PtF p0 = q[0], p1 = q[1], p2 = q[2], p3 = q[3];
                    nn_test<float, PtF>(p0, j, tx, ty, tz, best, bj, bo);
                    nn_test<float, PtF>(p1, j + 1, tx, ty, tz, best, bj, bo);
                    nn_test<float, PtF>(p2, j + 2, tx, ty, tz, best, bj, bo);
                    nn_test<float, PtF>(p3, j + 3, tx, ty, tz, best, bj, bo);
                }
#endif
            }
        }
        // certified: the ball of what the lane holds now lies inside the box that was just walked
        const bool has2 = best < a.bound2_f;
        const float r2 = has2 ? RealTraits<float>::sqrt_fast(best) * 1.000002f + g.slack : rmax;
        const BallBox b2 = ball_cells(g, tx, ty, tz, r2);
        const bool inside = b2.x0 >= X0 && b2.x1 <= X1 && b2.y0 >= Y0 && b2.y1 <= Y1 && b2.z0 >= Z0 && b2.z1 <= Z1;
        pending = pending && !inside;
    }
    if (pending) nn_search<float, PtF, false, true>(g, a.pts, a.cell_start, tx, ty, tz, a.bound2_f, best, bj, bo);
    if (exists) {
        const bool ok = live && bj != PCR_NONE && __builtin_sqrtf(best) < a.md_f;
        a.nn_j[i] = ok ? bj : PCR_NONE;
    }
}

template <int SEED>
__global__ void __launch_bounds__(256) k_nn_coop(const LinArgs a) {
    PoseK P;
    if (!load_pose<false>(a, P)) return;
#if PCR_COOP_LDS
    __shared__ __attribute__((aligned(16))) PtF stage_all[4][PCR_COOP_CAP];
    PtF *stage = stage_all[threadIdx.x >> 6];
#else
    PtF *stage = nullptr;
#endif
    if (a.sched_local) nn_tile_loop<1, 64>(a, [&](int64_t first, int64_t end) { nn_tile_coop<SEED>(a, P, stage, first, end); });
    else nn_tile_loop<0, 64>(a, [&](int64_t first, int64_t end) { nn_tile_coop<SEED>(a, P, stage, first, end); });
}

// work counters of the search (instrumentation; same traversal as k_nn_scan<0>): out[0..3] = per-lane
// sums of rings, rows loaded, rows pruned by arithmetic, candidates tested; out[4..7] = the same with
// the per-WAVE maximum charged to all 64 lanes (what the SIMD actually executes under divergence)
template <int HALO>
__global__ void __launch_bounds__(256) k_nn_counters(const LinArgs a, unsigned long long *out) {
    const TileIter it(a);
    unsigned long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long cyc[3] = {0, 0, 0};                  // wave wall-clock: prologue, ring 0, outer rings
    for (int64_t i0 = it.base - threadIdx.x; i0 < it.end; i0 += it.stride) {
        const int64_t i = i0 + threadIdx.x;
        NNStats st = {0, 0, 0, 0};
        const unsigned long long t0 = __builtin_readcyclecounter();
        float tx = 0, ty = 0, tz = 0;
        const bool live = i < it.end;
        if (live) {
            const float x = a.sx[i], y = a.sy[i], z = a.sz[i];
            xform(a.hp, x, y, z, tx, ty, tz);
        }
        uint32_t bj = PCR_NONE, bo = PCR_NONE; float best = a.bound2_f;
        NNCell<float> c = nn_cell<float>(a.gf, tx, ty, tz, a.bound2_f);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        const unsigned long long t1 = __builtin_readcyclecounter();
        int kstart = 0;
        if (live) kstart = nn_ring0<float, PtF, true, HALO != 0>(a.gf, a.pts, a.cell_start, c, tx, ty, tz, best, bj, bo, &st);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        const unsigned long long t2 = __builtin_readcyclecounter();
        if (live) nn_rings<float, PtF, true>(a.gf, a.pts, a.cell_start, c, kstart, tx, ty, tz, best, bj, bo, &st);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        const unsigned long long t3 = __builtin_readcyclecounter();
        cyc[0] += t1 - t0; cyc[1] += t2 - t1; cyc[2] += t3 - t2;
        uint32_t v[4] = {st.rings, st.rows_loaded, st.rows_pruned, st.cand};
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
            uint32_t m = v[c4];
            for (int off = 32; off >= 1; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off, 64));
            acc[c4] += v[c4];
            acc[4 + c4] += m;
        }
    }
    for (int c = 0; c < 8; ++c) atomicAdd(&out[c], acc[c]);
    if ((threadIdx.x & 63) == 0) for (int c = 0; c < 3; ++c) atomicAdd(&out[8 + c], cyc[c]);
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
    if (threadIdx.x == 0) {
        if (f.kind != PCR_ICP) {
            for (int i = 0; i < 29; ++i) f.out[i] = tot[i];
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
            for (int i = 0; i < 6; ++i) for (int j = i; j < 6; ++j) f.out[p++] = H[i][j];
            for (int i = 0; i < 3; ++i) { f.out[21 + i] = tot[10 + i]; f.out[24 + i] = tot[13 + i]; }
            f.out[27] = tot[16]; f.out[28] = cnt;
        }
        f.out[29] = mode == PCR_NN_LIST ? (double)listed : 0.0; f.out[30] = (double)mode; f.out[31] = 0;
        if (f.host_out) {
            for (int i = 0; i < 31; ++i) f.host_out[i] = f.out[i];
            __threadfence_system();
            *f.host_flag = f.seq;
        }
    }
}

// The O(1) tail of an iteration on the device (ONE thread): record the trace row, dx = -solve(H, g),
// |dx| < tol test, T <- plus(T, dx), derived float32 / rotation copies for the next pass, progress
// words for the host.  out29 is complete and visible to this thread.
__device__ __forceinline__ void gn_update(const FinArgs &f, double (*A)[7]) {
    PoseDev *p = f.pose;
    const int it = p->iter;
    double T[16];
    for (int i = 0; i < 16; ++i) T[i] = p->T[i];
    if (f.trace) {
        double *row = f.trace + (size_t)it * 45;
        for (int i = 0; i < 16; ++i) row[i] = T[i];
        for (int i = 0; i < 29; ++i) row[16 + i] = f.out[i];
    }
    const int r = gn_step(A, f.out, f.tol, T);
    int done = r == 2 ? PCR_LOOP_SINGULAR : (r == 1 ? PCR_LOOP_CONVERGED : PCR_LOOP_RUNNING);
    if (r == 0) {
        for (int i = 0; i < 16; ++i) p->T[i] = T[i];
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) { p->R[3 * i + j] = T[4 * i + j]; p->r32[3 * i + j] = (float)T[4 * i + j]; }
            p->t32[i] = (float)T[4 * i + 3];
        }
    }
    const int it1 = it + 1;
    if (done == PCR_LOOP_RUNNING && it1 >= f.max_iter) done = PCR_LOOP_MAXITER;
    p->iter = it1;
    p->done = done;
    if (f.host_T) {
        // progress word every iteration (the host keeps the queue two iterations ahead of it); the pose
        // itself crosses PCIe only once, with the final state
        if (done != PCR_LOOP_RUNNING) {
            for (int i = 0; i < 16; ++i) f.host_T[i] = T[i];
            __threadfence_system();
        }
        *f.host_state = ((unsigned long long)(unsigned)done << 32) | (unsigned)it1;
    }
}

// the step of the device-resident loop: a 1-wave launch behind the fold (single GPU) or behind the
// all-reduce of the 29 sums (multi-GPU: every rank computes the same update)
__global__ void __launch_bounds__(64) k_gn_update(const FinArgs f) {
    if (threadIdx.x == 0 && f.pose->done == PCR_LOOP_RUNNING) {
        double A[6][7];                 // registers: gn_solve6 indexes it with compile-time constants only
        gn_update(f, A);
    }
}

// start of pcr_align: the initial pose into HBM (kernel arguments: no host-to-device copy)
struct PoseInit { double T[16]; };
__global__ void __launch_bounds__(64) k_pose_init(PoseDev *p, const PoseInit init, int max_iter) {
    if (threadIdx.x != 0) return;
    for (int i = 0; i < 16; ++i) p->T[i] = init.T[i];
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) { p->R[3 * i + j] = init.T[4 * i + j]; p->r32[3 * i + j] = (float)init.T[4 * i + j]; }
        p->t32[i] = (float)init.T[4 * i + 3];
    }
    p->iter = 0;
    p->done = max_iter > 0 ? PCR_LOOP_RUNNING : PCR_LOOP_MAXITER;
}

// Stand-alone fold (variant 0, and PCR_FUSE_FINALIZE=0): ONE block of NT threads.
template <int NT>
__device__ __forceinline__ void finalize_body(const FinArgs &f) {
    __shared__ double part[32][33];
    __shared__ double tot[32];
    constexpr int RPT = 32 / (NT / 32);                        // row-groups per thread: 1 (1024 threads) or 4 (256)
    const int c = threadIdx.x & 31, r0 = threadIdx.x >> 5;     // 32 row-groups x 32 components
    // RPT x 8 independent loads in flight per thread (a single dependent chain is pure latency)
    double s8[RPT][8];
#pragma unroll
    for (int q = 0; q < RPT; ++q)
#pragma unroll
        for (int u = 0; u < 8; ++u) s8[q][u] = 0.0;
    for (int b00 = 0; b00 < f.nblocks; b00 += 256) {
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int b = b00 + r0 + q * (NT / 32) + 32 * u;
                double v = 0.0;
                if (b < f.nblocks) {
                    const double *src = &f.partials[(size_t)b * 32 + c];
                    v = *src;
                }
                s8[q][u] += v;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < RPT; ++q)
        part[r0 + q * (NT / 32)][c] = ((s8[q][0] + s8[q][1]) + (s8[q][2] + s8[q][3])) + ((s8[q][4] + s8[q][5]) + (s8[q][6] + s8[q][7]));
    __syncthreads();
    if (threadIdx.x < 32) {
        double t = 0.0;
        for (int k = 0; k < 32; ++k) t += part[k][threadIdx.x];
        tot[threadIdx.x] = t;
    }
    __syncthreads();
    finalize_emit(f, tot);
}

__global__ void __launch_bounds__(1024) k_finalize(const FinArgs f) {
    if (f.pose && f.pose->done != PCR_LOOP_RUNNING) return;
    finalize_body<1024>(f);
}

template <int KIND>
__global__ void __launch_bounds__(256) k_reduce(const LinArgs a) {
    PoseK P;
    if (!load_pose<true>(a, P)) return;
    double acc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = 0.0;
    const TileIter it(a);
    reduce_stream<KIND>(acc, a, P, it.base, it.end, it.stride);
    block_store_partials(acc, a.partials);
}

// The fold of the per-block partial sums INSIDE the producing kernel (no separate k_finalize launch: ~10 us
// and a launch gap per pass).  Any grid that is a multiple of 8 blocks; two levels of tickets.  Blocks g,
// g+8, g+16, ... form group g (the blocks the dispatcher places on XCD g, so a group's traffic stays in one
// L2 -- a locality assumption only, every cross-block access is coherent at agent scope).  The block that
// takes a group's last ticket folds the group's partials into row nblocks+g; the group leader that takes the
// last of the 8 second-level tickets folds those rows and emits.  8 x (nblocks/8) + 8 serialised atomics
// instead of nblocks.
__device__ __forceinline__ void ticket_fold_emit(double *acc, const LinArgs &a, const FinArgs &f) {
    block_store_partials<true>(acc, a.partials);

    __shared__ int role;
    __shared__ double part[8][33];
    __shared__ double tot[32];
    const int ng = 8;          // (a single group for small grids was measured: no gain, 15.9 vs 15.7 us at 100 k points)
    const int g = (int)(blockIdx.x & 7), per = f.nblocks / ng;
    uint32_t *ctr1 = &f.tickets[g * 16], *ctr2 = &f.tickets[8 * 16];
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
    if (!role) return;

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
    if (!role) return;

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
}

// k_reduce with the fold inside (the shipped reduce kernel)
template <int KIND>
__global__ void __launch_bounds__(256) k_reduce_finalize(const LinArgs a, const FinArgs f) {
    PoseK P;
    if (!load_pose<true>(a, P)) return;
    double acc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = 0.0;
    const TileIter it(a);
    reduce_stream<KIND>(acc, a, P, it.base, it.end, it.stride);
    ticket_fold_emit(acc, a, f);
    // (the Gauss-Newton step is NOT inlined here: its straight-line float64 code needs 136 VGPRs, which
    // would cap this streaming kernel at 3 blocks per CU; k_gn_update runs it as a 1-wave launch)
}

template <int KIND, int HALO>
__global__ void __launch_bounds__(256) k_linearize_finalize(const LinArgs a, const FinArgs f) {
    PoseK P;
    if (!load_pose<true>(a, P)) return;
    double acc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = 0.0;
    linearize_body<KIND, HALO>(a, P, acc);
    ticket_fold_emit(acc, a, f);
}

// after the RCCL all-reduce: hand the 29 doubles to the host the same zero-copy way k_finalize does
__global__ void __launch_bounds__(64) k_publish(const double *__restrict__ out, double *host_out,
                                                volatile uint32_t *host_flag, uint32_t seq) {
    if (threadIdx.x < 29) host_out[threadIdx.x] = out[threadIdx.x];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) *host_flag = seq;
}

// ---- fine seam: plain NN queries (no transform), original indices out -------------------------
template <typename Real, typename PT, bool HALO>
__global__ void __launch_bounds__(256) k_nn_query(Geom<Real> g, const PT *pts, const uint32_t *cs,
                                                  const float *q, int64_t m, Real bound2, Real rmax,
                                                  Real *dist, int64_t *idx) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    Real best; uint32_t bj, bo;
    nn_search<Real, PT, false, false, HALO>(g, pts, cs, (Real)q[3 * i], (Real)q[3 * i + 1], (Real)q[3 * i + 2], bound2, best, bj, bo);
    bj = bo;                                       // (only tested against PCR_NONE below)
    Real d = RealTraits<Real>::sqrt_rn(best);
    if (bj != PCR_NONE && rmax < RealTraits<Real>::inf() && !(d < rmax)) bj = PCR_NONE;
    dist[i] = bj == PCR_NONE ? RealTraits<Real>::inf() : d;
    idx[i] = bj == PCR_NONE ? (int64_t)-1 : (int64_t)bo;
}

// =============================================================================================
// host side
// =============================================================================================
pcr_status pcr_ensure_scratch(pcr_context *ctx, int64_t n_points) {
    (void)n_points;
    if (!ctx->d_partials) {
        ctx->max_blocks = (ctx->num_cu * 16 + 7) & ~7;
        // + 8 rows: the group sums of k_reduce_finalize live behind the per-block rows
        HIP_TRY(pcr_malloc_retry((void **)&ctx->d_partials, sizeof(double) * 32 * (size_t)(ctx->max_blocks + 8)));
        HIP_TRY(pcr_malloc_retry((void **)&ctx->d_out, sizeof(double) * 32));
        HIP_TRY(pcr_malloc_retry((void **)&ctx->d_pose, sizeof(PoseDev)));
        // pinned + mapped: [0..28] sums, [32] sequence number (pcr_linearize); [40..55] pose, [56] loop state (pcr_align)
        HIP_TRY(hipHostMalloc(&ctx->h_out, sizeof(double) * 64, hipHostMallocMapped | hipHostMallocCoherent));
        memset(ctx->h_out, 0, sizeof(double) * 64);
        HIP_TRY(hipHostGetDevicePointer((void **)&ctx->h_out_dev, ctx->h_out, 0));
        // 8 + 1 tickets (64 B apart), then the tile counters
        const size_t ctr_words = 9 * 16 + (size_t)PCR_TILE_CTRS * PCR_TILE_STRIDE;
        HIP_TRY(pcr_malloc_retry((void **)&ctx->d_tile_ctr, sizeof(uint32_t) * ctr_words));
        HIP_TRY(hipMemsetAsync(ctx->d_tile_ctr, 0, sizeof(uint32_t) * ctr_words, ctx->stream));
        for (int v = 0; v < 3; ++v) {
            int nb = 0;
            hipError_t e = v == 0 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_nn_scan<0, 0, 1, 0>, 256, 0)
                         : v == 1 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_nn_scan<1, 0, 0, 0>, 256, 0)
                                  : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_nn_coop<1>, 256, 0);
            ctx->nn_blocks_per_cu[v] = (e == hipSuccess && nb > 0) ? nb : 4;
        }
    }
    return PCR_OK;
}

static int choose_blocks(const pcr_context *ctx, int64_t n) {
    // enough 256-thread blocks to fill every CU several times over, never more than the work,
    // always a multiple of 8 (one contiguous span of the scan per XCD; odd CU counts round down)
    int64_t want = (n + 255) / 256;
    int64_t cap = (int64_t)ctx->num_cu * 8;
    int64_t nb = want < cap ? want : cap;
    nb = (nb + 7) & ~(int64_t)7;
    if (nb > ctx->max_blocks) nb = ctx->max_blocks;
    nb &= ~(int64_t)7;
    if (nb < 8) nb = 8;
    return (int)nb;
}

// ---- one pass = NN kernel + reduce kernel (variant 1) or the fused kernel (variant 0) ------------
struct Pass {
    pcr_context *ctx;
    pcr_target *t;
    pcr_scan *s;
    int kind;
    LinArgs a;
    FinArgs f;
    bool one_kernel;     // fused search + reduce kernel (variant 0, or variant 2 on a small scan)
    bool fused_fin;      // the fold of the block partials inside the producing kernel instead of k_finalize
    int nn_mode;         // PCR_NN_FULL / TRACK / LIST
    bool reuse_ready;    // the scan has the buffers of the certified-reuse path
};

static pcr_status pass_setup(Pass *ps, pcr_target *t, pcr_scan *s, int kind, double max_dist, unsigned flags) {
    pcr_context *ctx = t->ctx;
    PCR_REQUIRE(s->ctx == ctx, "scan and target belong to different contexts");
    PCR_REQUIRE(kind >= PCR_ICP && kind <= PCR_NDT, "unknown kind");
    PCR_REQUIRE(max_dist > 0, "max_dist must be positive");
    if ((kind == PCR_ICP || kind == PCR_PLANE) && t->is_voxel) {
        pcr_set_error("kind %d needs a point target", kind);
        return PCR_ERR_NO_TARGET;
    }
    if ((kind == PCR_VPLANE || kind == PCR_NDT) && !t->is_voxel) {
        pcr_set_error("kind %d needs a voxel target", kind);
        return PCR_ERR_NO_TARGET;
    }
    if (kind == PCR_PLANE && !t->pn) { pcr_set_error("PlaneICP target has no normals"); return PCR_ERR_NO_TARGET; }
    if (kind == PCR_VPLANE && !t->vnorm) { pcr_set_error("VPlaneICP target has no voxel normals"); return PCR_ERR_NO_TARGET; }
    if (kind == PCR_NDT && !t->vicov) { pcr_set_error("NDT target has no inverse covariances"); return PCR_ERR_NO_TARGET; }
    HIP_TRY(hipSetDevice(ctx->device));
    PCR_TRY(pcr_ensure_scratch(ctx, s->n));
    // one fused kernel or search + reduce?  variant 2 (default) decides by size: a small scan is latency-bound
    // and runs fused (tools/variant_crossover.py: 100 k points 74 vs 80 us per pass, 300 k 97 vs 92, 1.06 M 190 vs 151)
    bool one_kernel = ctx->variant == 0;
    if (ctx->variant == 2) one_kernel = s->n <= (int64_t)ctx->num_cu * 1024;   // measured crossover: 200 k - 300 k points on 256 CUs
    if (!one_kernel && !s->nn_j) {
        HIP_TRY(pcr_malloc_retry((void **)&s->nn_j, sizeof(uint32_t) * (size_t)(s->n > 0 ? s->n : 1)));
        s->nn_serial = 0;
    }
    const int nblocks_split = [&] {
        int nb = choose_blocks(ctx, s->n);
        if (nb > ctx->num_cu * 4) nb = ctx->num_cu * 4;
        nb &= ~7;
        return nb < 8 ? 8 : nb;
    }();
    ps->reuse_ready = false;
    if (!one_kernel && ctx->reuse != 0 && ctx->nn_mode == 0 && s->n > 0) {
        if (!s->lb2) {
            const size_t words = (((size_t)s->n + 63) / 64 + 31) & ~(size_t)15;     // whole 16-word chunks + slack
            HIP_TRY(pcr_malloc_retry((void **)&s->lb2, sizeof(float) * (size_t)s->n));
            HIP_TRY(pcr_malloc_retry((void **)&s->umask, sizeof(unsigned long long) * words));
            HIP_TRY(hipMemsetAsync(s->umask, 0, sizeof(unsigned long long) * words, ctx->stream));
            s->track_valid = false;
        }
        if (s->ucnt_cap < nblocks_split) {
            if (s->ucnt) HIP_TRY(hipFree(s->ucnt));
            s->ucnt = nullptr; s->ucnt_cap = 0;
            HIP_TRY(pcr_malloc_retry((void **)&s->ucnt, sizeof(uint32_t) * (size_t)nblocks_split));
            s->ucnt_cap = nblocks_split;
        }
        ps->reuse_ready = true;
    }
    ps->ctx = ctx; ps->t = t; ps->s = s; ps->kind = kind; ps->one_kernel = one_kernel;
    LinArgs &a = ps->a;
    memset(&a, 0, sizeof a);
    a.sx = s->x; a.sy = s->y; a.sz = s->z; a.n = s->n;
    a.gf = t->gf; a.pts = t->pts; a.pn = t->pn;
    a.gd = t->gd; a.means = t->means; a.vnorm = t->vnorm; a.vicov = t->vicov;
    a.cell_start = t->cell_start;
    a.md_f = (float)max_dist; a.md_d = max_dist;
    const double bound = max_dist * (1.0 + 1e-6);
    a.bound2_f = (float)(bound * bound); a.bound2_d = bound * bound;
    a.flags = flags;
    a.nblocks = choose_blocks(ctx, s->n);
    a.partials = ctx->d_partials;
    a.nn_j = s->nn_j; a.tile_ctr = ctx->d_tile_ctr + 9 * 16;
    a.lb2 = s->lb2; a.umask = s->umask; a.ucnt = s->ucnt;
    a.mu_f = (float)(ctx->reuse_mu * (t->is_voxel ? t->gd.h : (double)t->gf.h));
    if (!one_kernel && a.nblocks > ctx->num_cu * 4) a.nblocks = ctx->num_cu * 4;   // k_reduce streams: 4 blocks/CU
    // TileIter and the ticket counts of k_reduce_finalize need a multiple of 8 blocks
    a.nblocks &= ~7;
    if (a.nblocks < 8) a.nblocks = 8;
    ps->fused_fin = ctx->fuse_finalize;
    ps->nn_mode = PCR_NN_FULL;
    FinArgs &f = ps->f;
    memset(&f, 0, sizeof f);
    f.ucnt = s->ucnt; f.n_ucnt = a.nblocks;
    f.partials = ctx->d_partials; f.tile_ctr = ctx->d_tile_ctr + 9 * 16; f.tickets = ctx->d_tile_ctr; f.nblocks = a.nblocks; f.kind = kind; f.out = ctx->d_out;
    return PCR_OK;
}

static void pass_set_host_pose(Pass *ps, const double T[16]) {
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) {
            ps->a.hp.R[3 * i + j] = T[4 * i + j]; ps->a.hp.r32[3 * i + j] = (float)T[4 * i + j];
            ps->f.R[3 * i + j] = T[4 * i + j];
        }
        ps->a.hp.t32[i] = (float)T[4 * i + 3];
    }
    ps->a.pose = nullptr; ps->f.pose = nullptr;
    if (ps->s->pose_valid) {
        const double *Tp = ps->s->prev_T;
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) ps->a.hq.r32[3 * i + j] = (float)Tp[4 * i + j];
            ps->a.hq.t32[i] = (float)Tp[4 * i + 3];
        }
    }
}

// What the search of a pass does (gn_math.h: gn_choose_nn_mode) and the search bound that goes with it: a tracking
// search looks 5 % beyond the gate, so that a point with nothing in reach can be certified "still nothing" later.
static void pass_set_mode(Pass *ps, int mode) {
    ps->nn_mode = mode;
    ps->f.nn_mode = mode;
    const double md = ps->a.md_d;
    const double bound = mode == PCR_NN_FULL ? md * (1.0 + 1e-6) : md * 1.05;
    ps->a.bound2_f = (float)(bound * bound); ps->a.bound2_d = bound * bound;
}

static int host_choose_mode(const Pass *ps, const double T[16], double *motion_out) {
    const pcr_context *ctx = ps->ctx;
    const pcr_scan *s = ps->s;
    const pcr_target *t = ps->t;
    *motion_out = -1.0;
    if (!ps->reuse_ready) return PCR_NN_FULL;
    const int have_prev = s->pose_valid && s->nn_serial == t->serial && s->nn_serial != 0;
    if (!have_prev) return PCR_NN_FULL;
    const double h = t->is_voxel ? t->gd.h : (double)t->gf.h;
    const double m = gn_typical_motion(s->prev_T, T, s->bb_c, s->bb_e);
    *motion_out = m;
    return gn_choose_nn_mode(ctx->reuse, have_prev, s->track_valid ? 1 : 0, m, s->last_motion, ctx->reuse_tau * h);
}

template <int KIND>
static void launch_reduce_kind(const Pass *ps, bool fused, dim3 grid) {
    if (fused) hipLaunchKernelGGL(k_reduce_finalize<KIND>, grid, dim3(256), 0, ps->ctx->stream, ps->a, ps->f);
    else hipLaunchKernelGGL(k_reduce<KIND>, grid, dim3(256), 0, ps->ctx->stream, ps->a);
}

// enqueue the kernels of one pass on the context's stream (no waiting)
static pcr_status pass_enqueue(Pass *ps) {
    pcr_context *ctx = ps->ctx;
    const LinArgs &a = ps->a;
    const dim3 grid(a.nblocks), block(256);
    ProfEvent ev;
    if (ctx->prof_on) ctx->prof_this_pass = (ctx->prof_pass++ % (uint64_t)ctx->prof_period) == 0;
    if (ps->one_kernel) {
        pcr_prof_begin(ctx, PCR_K_LINEARIZE, &ev);
        RoctxRange range("pcr:linearize");
        const bool halo = !ps->t->is_voxel && ps->t->cs_h != nullptr;
#define PCR_LIN_CASE(K)                                                                                         \
        if (ps->fused_fin) {                                                                                    \
            if (halo) hipLaunchKernelGGL((k_linearize_finalize<K, 1>), grid, block, 0, ctx->stream, a, ps->f);  \
            else hipLaunchKernelGGL((k_linearize_finalize<K, 0>), grid, block, 0, ctx->stream, a, ps->f);       \
        } else {                                                                                                \
            if (halo) hipLaunchKernelGGL((k_linearize<K, 1>), grid, block, 0, ctx->stream, a);                  \
            else hipLaunchKernelGGL((k_linearize<K, 0>), grid, block, 0, ctx->stream, a);                       \
        }
        switch (ps->kind) {
        case PCR_ICP: PCR_LIN_CASE(PCR_ICP) break;
        case PCR_PLANE: PCR_LIN_CASE(PCR_PLANE) break;
        case PCR_VPLANE: PCR_LIN_CASE(PCR_VPLANE) break;
        default: PCR_LIN_CASE(PCR_NDT) break;
        }
#undef PCR_LIN_CASE
        pcr_prof_end(ctx, &ev);
    } else {
        const bool vox = ps->t->is_voxel != 0;
        const int mode = ps->nn_mode;
        if (mode == PCR_NN_LIST) {
            // the previous matches that are provably still exact need no search (k_certify)
            pcr_prof_begin(ctx, PCR_K_CERTIFY, &ev);
            RoctxRange range("pcr:certify");
            if (vox) hipLaunchKernelGGL(k_certify<1>, grid, block, 0, ctx->stream, a);
            else hipLaunchKernelGGL(k_certify<0>, grid, block, 0, ctx->stream, a);
            pcr_prof_end(ctx, &ev);
        }
        pcr_prof_begin(ctx, PCR_K_NN, &ev);
        {   // exactly one resident generation of waves; they share the tiles dynamically
            RoctxRange range("pcr:nn_search");
            int64_t nb = (int64_t)ctx->num_cu * ctx->nn_blocks_per_cu[vox ? 1 : (ctx->nn_mode == 2 ? 2 : 0)];
            // tiles of the hand-out: 64 points per wave, or the 1024-point chunks of a LIST pass (one block per chunk)
            const int64_t tiles = mode == PCR_NN_LIST ? (a.n + PCR_LIST_CHUNK - 1) / PCR_LIST_CHUNK : (a.n + 63) / 64;
            const int64_t need = mode == PCR_NN_LIST ? tiles : (tiles + 3) / 4;
            if (nb > need) nb = need;
            nb = (nb + 7) & ~(int64_t)7;
            if (nb < 8) nb = 8;
            const dim3 nn_grid((unsigned)nb);
            // hand-out policy: at most ~1.5 tiles per launched wave -> block-local (nn_tile_loop)
            ps->a.sched_local = ctx->tile_local >= 0 ? ctx->tile_local : (tiles * 2 <= nb * 4 * 3 ? 1 : 0);
            if (!vox && ctx->nn_mode == 2) {
                hipLaunchKernelGGL((k_nn_coop<0>), nn_grid, block, 0, ctx->stream, a);
            } else if (!vox) {
                launch_nn_scan<0>(mode, ps->t->cs_h != nullptr, ps->a.sched_local != 0, nn_grid, ctx->stream, a);
            } else {
                launch_nn_scan<1>(mode, false, ps->a.sched_local != 0, nn_grid, ctx->stream, a);
            }
            ps->s->nn_serial = ps->t->serial;      // nn_j now holds matches against this target
        }
        pcr_prof_end(ctx, &ev);
        pcr_prof_begin(ctx, PCR_K_REDUCE, &ev);
        RoctxRange range("pcr:reduce");
        switch (ps->kind) {
        case PCR_ICP: launch_reduce_kind<PCR_ICP>(ps, ps->fused_fin, grid); break;
        case PCR_PLANE: launch_reduce_kind<PCR_PLANE>(ps, ps->fused_fin, grid); break;
        case PCR_VPLANE: launch_reduce_kind<PCR_VPLANE>(ps, ps->fused_fin, grid); break;
        default: launch_reduce_kind<PCR_NDT>(ps, ps->fused_fin, grid); break;
        }
        pcr_prof_end(ctx, &ev);
    }
    HIP_TRY(hipGetLastError());
    if (!ps->fused_fin) {
        pcr_prof_begin(ctx, PCR_K_FINALIZE, &ev);
        hipLaunchKernelGGL(k_finalize, dim3(1), dim3(1024), 0, ctx->stream, ps->f);
        pcr_prof_end(ctx, &ev);
        HIP_TRY(hipGetLastError());
    }
    return PCR_OK;
}

// Spin on a word in pinned host memory that a kernel writes, then fall back to a blocking wait.
template <typename Pred>
static pcr_status wait_host_word(pcr_context *ctx, Pred ready, const char *what) {
    for (long spin = 0; spin < 4000000L; ++spin) {
        if (ready()) { __atomic_thread_fence(__ATOMIC_ACQUIRE); return PCR_OK; }
        __builtin_ia32_pause();
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));     // very long pass (or a fault): block, then re-check
    if (!ready()) { pcr_set_error("%s did not report completion", what); return PCR_ERR_HIP; }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    return PCR_OK;
}

// Rare 6-60 ms stalls of ONE call early in a process ("12-42 ms, about once per 1500 passes") were
// root-caused in round 3 (tools/stall_study.py, PCR_STALL_DEBUG=1; profiles/r03_stall_root_cause.txt): the calling
// thread is DESCHEDULED -- on a CPU for 0.03-0.08 ms of a 41 ms stall -- while the container's cgroup reports one
// more throttled period (cpu.max = 16 CPUs per 100 ms on the GPU box): the thread pools of the host libraries
// (256 visible CPUs) burn the CPU quota during start-up and the kernel parks every thread of the cgroup until the
// next period.  Durations are 6.6 ms + k x 10 ms (scheduler ticks); the stall lands in the enqueue or in the spin,
// wherever the thread happens to be.  Neither the GPU nor the HIP runtime is involved: with the host thread pools
// capped (OMP_NUM_THREADS / OPENBLAS_NUM_THREADS) 25 of 25 fresh processes and 1e6 consecutive passes stay below
// 1 ms.  The hipStreamQuery cadence of round 2 rested on a wrong theory (20 of 25 processes stalled with it, 23 of
// 25 without); it is kept only as an opt-in knob (PCR_RETIRE_PERIOD, default off).
static void retire_completed(pcr_context *ctx) {
    if (ctx->retire_period > 0 && ++ctx->passes_since_query >= (uint32_t)ctx->retire_period) {
        ctx->passes_since_query = 0;
        (void)hipStreamQuery(ctx->stream);
    }
}

pcr_status pcr_run_linearize(pcr_target *t, pcr_scan *s, int kind, const double T[16], double max_dist,
                             unsigned flags, double out[29]) {
    Pass ps;
    PCR_TRY(pass_setup(&ps, t, s, kind, max_dist, flags));
    pcr_context *ctx = ps.ctx;
    pass_set_host_pose(&ps, T);
    double motion = -1.0;
    const int mode = ps.one_kernel ? PCR_NN_FULL : host_choose_mode(&ps, T, &motion);
    pass_set_mode(&ps, mode);
    const bool use_comm = ctx->comm != nullptr && !(flags & PCR_FLAG_LOCAL_ONLY);
    // single GPU: the finalize step writes the result and a sequence number straight into pinned host
    // memory (no copy command, no stream query); with a communicator the all-reduce sits in between
    const bool direct = !use_comm && ctx->h_out_dev != nullptr;
    volatile uint32_t *flag = (volatile uint32_t *)(ctx->h_out + 32);
    ps.f.host_out = direct ? ctx->h_out_dev : nullptr;
    ps.f.host_flag = direct ? (volatile uint32_t *)(ctx->h_out_dev + 32) : nullptr;
    const uint32_t seq = ++ctx->seq;
    ps.f.seq = seq;
    const bool stall_dbg = ctx->stall_debug;
    struct timespec ts0, ts1, ts2, tc0, tc2;
    if (stall_dbg) { clock_gettime(CLOCK_MONOTONIC, &ts0); clock_gettime(CLOCK_THREAD_CPUTIME_ID, &tc0); }
    PCR_TRY(pass_enqueue(&ps));
    if (stall_dbg) clock_gettime(CLOCK_MONOTONIC, &ts1);
    // what this pass leaves behind for the next one over the same scan
    memcpy(s->prev_T, T, sizeof s->prev_T);
    s->pose_valid = !ps.one_kernel;
    s->track_valid = mode != PCR_NN_FULL;
    s->last_mode = mode; s->last_motion = motion; s->last_marked = mode == PCR_NN_LIST ? -1 : 0;
    s->st_passes[mode] += 1;

    bool flagged = direct;
    if (use_comm) {
        ProfEvent ev;
        pcr_prof_begin(ctx, PCR_K_ALLREDUCE, &ev);
        RoctxRange range("pcr:allreduce29");
        pcr_status cs = pcr_comm_allreduce29(ctx, ctx->d_out);
        if (cs == PCR_OK && ctx->h_out_dev) {
            hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, ctx->stream, ctx->d_out, ctx->h_out_dev,
                               (volatile uint32_t *)(ctx->h_out_dev + 32), seq);
            flagged = true;
        }
        pcr_prof_end(ctx, &ev);
        if (cs != PCR_OK) return cs;
        HIP_TRY(hipGetLastError());
    }
    if (flagged) {
        PCR_TRY(wait_host_word(ctx, [&] { return *flag == seq; }, "finalize kernel"));
        if (stall_dbg) {                     // developer (PCR_STALL_DEBUG): where did a slow call spend its time?
            clock_gettime(CLOCK_MONOTONIC, &ts2); clock_gettime(CLOCK_THREAD_CPUTIME_ID, &tc2);
            const double cpu = (tc2.tv_sec - tc0.tv_sec) * 1e3 + (tc2.tv_nsec - tc0.tv_nsec) * 1e-6;
            const double enq = (ts1.tv_sec - ts0.tv_sec) * 1e3 + (ts1.tv_nsec - ts0.tv_nsec) * 1e-6;
            const double wait = (ts2.tv_sec - ts1.tv_sec) * 1e3 + (ts2.tv_nsec - ts1.tv_nsec) * 1e-6;
            if (enq + wait > 1.0) {
                fprintf(stderr, "[pcr stall] seq %u: enqueue %.3f ms, wait for the result %.3f ms; this thread was ON a CPU for %.3f ms of it\n",
                        seq, enq, wait, cpu);
            }
        }
        for (int i = 0; i < 29; ++i) out[i] = ctx->h_out[i];
        if (mode == PCR_NN_LIST && direct) {
            s->last_marked = (int64_t)ctx->h_out[29];
            s->st_marked += s->last_marked; s->st_listed_of += s->n;
        }
        retire_completed(ctx);
        return PCR_OK;
    }
    HIP_TRY(hipMemcpyAsync(ctx->h_out, ctx->d_out, sizeof(double) * 31, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < 29; ++i) out[i] = ctx->h_out[i];
    if (mode == PCR_NN_LIST && !use_comm) {
        s->last_marked = (int64_t)ctx->h_out[29];
        s->st_marked += s->last_marked; s->st_listed_of += s->n;
    }
    return PCR_OK;
}

// ---- Registration.align behind the boundary, device-resident (registration.py:71-113) -------------
// The pose lives in HBM; every iteration is k_nn_scan + k_reduce_finalize + k_gn_update (one wave:
// dx = -solve(H, g), the |dx| < tol test and T <- plus(T, dx)).  The host only keeps the
// queue a couple of iterations ahead of the GPU and watches two words in pinned memory: no host round
// trip, no device-to-host copy and no host solve between iterations.  Launches that arrive after
// convergence see pose->done and return at once.  With a communicator the 29 sums are all-reduced
// between the fold and the step (k_gn_update); every rank issues the same number of collectives (see below).
pcr_status pcr_run_align(pcr_target *t, pcr_scan *s, int kind, const double T_init[16], int max_iter, double tol,
                         double max_dist, unsigned flags, double T_out[16], int *iterations, double *trace_or_null) {
    Pass ps;
    PCR_TRY(pass_setup(&ps, t, s, kind, max_dist, flags));
    pcr_context *ctx = ps.ctx;
    if (max_iter <= 0) {
        memcpy(T_out, T_init, 16 * sizeof(double));
        if (iterations) *iterations = 0;
        return PCR_OK;
    }
    s->pose_valid = false; s->track_valid = false;       // (the loop's own bookkeeping lives in PoseDev)
    if (ctx->trace_cap < max_iter) {
        if (ctx->d_trace) HIP_TRY(hipFree(ctx->d_trace));
        ctx->d_trace = nullptr; ctx->trace_cap = 0;
        HIP_TRY(pcr_malloc_retry((void **)&ctx->d_trace, sizeof(double) * 45 * (size_t)max_iter));
        ctx->trace_cap = max_iter;
    }
    const bool use_comm = ctx->comm != nullptr && !(flags & PCR_FLAG_LOCAL_ONLY);
    volatile unsigned long long *state = (volatile unsigned long long *)(ctx->h_out + 56);
    *state = 0;
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
    ps.a.pose = ctx->d_pose;
    FinArgs &f = ps.f;
    f.pose = ctx->d_pose; f.max_iter = max_iter; f.tol = tol;
    f.trace = ctx->d_trace;
    f.host_T = ctx->h_out_dev + 40;
    f.host_state = (volatile unsigned long long *)(ctx->h_out_dev + 56);
    PoseInit init;
    memcpy(init.T, T_init, sizeof init.T);
    hipLaunchKernelGGL(k_pose_init, dim3(1), dim3(64), 0, ctx->stream, ctx->d_pose, init, max_iter);
    HIP_TRY(hipGetLastError());

    auto passes_done = [&] { return (int)(unsigned)(*state & 0xffffffffull); };
    auto loop_done = [&] { return (int)(unsigned)(*state >> 32); };
    // One iteration = the pass, (multi-GPU) the in-stream all-reduce of its 29 sums, the one-wave update.  The host
    // keeps the queue AHEAD iterations beyond the one the GPU reports and never waits on the stream.
    // Multi-GPU: every rank must issue the SAME number of collectives, but each reads the (identical, all-reduced)
    // state word at its own time.  A rank enqueues iteration e only while e < passes_done + AHEAD, so when it sees
    // the loop end after `it` passes it has enqueued at most min(max_iter, it + AHEAD) iterations -- a number every
    // rank can compute; each tops its queue up to exactly that (launches behind the end are no-ops, their
    // all-reduces move stale sums): no host synchronisation between iterations, at most AHEAD dead all-reduces.
    const int AHEAD = 2;
    int enq = 0;
    auto enqueue_iteration = [&]() -> pcr_status {
        retire_completed(ctx);
        PCR_TRY(pass_enqueue(&ps));
        if (use_comm) {
            ProfEvent ev;
            pcr_prof_begin(ctx, PCR_K_ALLREDUCE, &ev);
            pcr_status cs = pcr_comm_allreduce29(ctx, ctx->d_out);
            pcr_prof_end(ctx, &ev);
            if (cs != PCR_OK) return cs;
        }
        hipLaunchKernelGGL(k_gn_update, dim3(1), dim3(64), 0, ctx->stream, f);
        HIP_TRY(hipGetLastError());
        ++enq;
        return PCR_OK;
    };
    long spin = 0;
    for (;;) {
        if (loop_done() != PCR_LOOP_RUNNING) break;
        const int fin = passes_done();
        if (enq < max_iter && enq < fin + AHEAD) {
            PCR_TRY(enqueue_iteration());
            spin = 0;
            continue;
        }
        __builtin_ia32_pause();
        if (++spin > 4000000L) {    // a very long pass (or a fault): block, then look again
            HIP_TRY(hipStreamSynchronize(ctx->stream));
            if (loop_done() == PCR_LOOP_RUNNING && passes_done() == fin && !(enq < max_iter)) {
                pcr_set_error("device Gauss-Newton loop made no progress");
                return PCR_ERR_HIP;
            }
            spin = 0;
        }
    }
    if (use_comm) {
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        const int it_end = passes_done();
        const int target = it_end + AHEAD < max_iter ? it_end + AHEAD : max_iter;
        while (enq < target) PCR_TRY(enqueue_iteration());
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    const int done = loop_done(), it = passes_done();
    for (int i = 0; i < 16; ++i) T_out[i] = ctx->h_out[40 + i];
    if (iterations) *iterations = it;
    if (trace_or_null && it > 0) {
        HIP_TRY(hipMemcpyAsync(trace_or_null, ctx->d_trace, sizeof(double) * 45 * (size_t)it, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    if (done == PCR_LOOP_SINGULAR) {
        pcr_set_error("Singular matrix");
        return PCR_ERR_SINGULAR;
    }
    return PCR_OK;
}

pcr_status pcr_run_nn(pcr_target *t, const float *d_q, int64_t m, double r_max, void *d_dist, int64_t *d_idx, int f64) {
    pcr_context *ctx = t->ctx;
    if (m == 0) return PCR_OK;
    const dim3 grid((unsigned)((m + 255) / 256)), block(256);
    const bool bounded = r_max > 0 && r_max < 1e300 * 1e300;
    ProfEvent ev;
    ctx->prof_this_pass = ctx->prof_on;            // (pass sampling applies to passes only)
    pcr_prof_begin(ctx, PCR_K_NN, &ev);
    if (!f64) {
        PCR_REQUIRE(!t->is_voxel, "pcr_nn_query needs a point target (use pcr_nn_query_f64 for voxels)");
        const float inf = __builtin_inff();
        const double b = r_max * (1.0 + 1e-6);
        const float bound2 = bounded ? (float)(b * b) : inf;
        if (t->cs_h)
            hipLaunchKernelGGL((k_nn_query<float, PtF, true>), grid, block, 0, ctx->stream, t->gf, t->pts, t->cell_start, d_q, m,
                               bound2, bounded ? (float)r_max : inf, (float *)d_dist, d_idx);
        else
            hipLaunchKernelGGL((k_nn_query<float, PtF, false>), grid, block, 0, ctx->stream, t->gf, t->pts, t->cell_start, d_q, m,
                               bound2, bounded ? (float)r_max : inf, (float *)d_dist, d_idx);
    } else {
        PCR_REQUIRE(t->is_voxel, "pcr_nn_query_f64 needs a voxel target");
        const double inf = __builtin_inf();
        const double b = r_max * (1.0 + 1e-6);
        hipLaunchKernelGGL((k_nn_query<double, PtD, false>), grid, block, 0, ctx->stream, t->gd, t->means, t->cell_start, d_q, m,
                           bounded ? b * b : inf, bounded ? r_max : inf, (double *)d_dist, d_idx);
    }
    pcr_prof_end(ctx, &ev);
    HIP_TRY(hipGetLastError());
    return PCR_OK;
}

// ---- instrumentation: search work counters for one pose (point targets) ------------------------
extern "C" pcr_status pcr_nn_counters(pcr_target *t, pcr_scan *s, const double T[16], double max_dist, double out[11]) {
    PCR_REQUIRE(t && s && T && out, "NULL argument");
    PCR_REQUIRE(!t->is_voxel, "counters are implemented for point targets");
    pcr_context *ctx = t->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    PCR_TRY(pcr_ensure_scratch(ctx, s->n));
    LinArgs a;
    memset(&a, 0, sizeof a);
    a.sx = s->x; a.sy = s->y; a.sz = s->z; a.n = s->n;
    a.gf = t->gf; a.pts = t->pts; a.cell_start = t->cell_start;
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) a.hp.r32[3 * i + j] = (float)T[4 * i + j];
        a.hp.t32[i] = (float)T[4 * i + 3];
    }
    const double bound = max_dist * (1.0 + 1e-6);
    a.bound2_f = (float)(bound * bound);
    a.nblocks = choose_blocks(ctx, s->n);
    unsigned long long h[11];
    CtxScope scope(ctx);
    DevBuf<unsigned long long> d;
    HIP_TRY(d.alloc(11));
    HIP_TRY(hipMemsetAsync(d.p, 0, sizeof h, ctx->stream));
    if (t->cs_h) hipLaunchKernelGGL(k_nn_counters<1>, dim3(a.nblocks), dim3(256), 0, ctx->stream, a, d.p);
    else hipLaunchKernelGGL(k_nn_counters<0>, dim3(a.nblocks), dim3(256), 0, ctx->stream, a, d.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(h, d.p, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < 11; ++i) out[i] = (double)h[i];
    return PCR_OK;
}
