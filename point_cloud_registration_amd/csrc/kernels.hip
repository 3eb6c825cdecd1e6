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
#include "pass_device.h"
#ifdef PCR_DEV
#include "nn_mfma.h"      // (round 5: the MFMA-filtered search, measured slower than k_nn_scan at every pose: developer build only)
#endif

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
    for (uint32_t e = threadIdx.x; e < total; e += 256) nn_point<VOXEL, HALO, 1>(a, a.gf, P, Q, first + (int64_t)lst[e]);
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
// VOXEL: 0 point target (float32 search), 1 centroid search with the plain row loop, 2 centroid search that takes the rows
// of a ring from the row-occupancy bitmap -- chosen per launch when the gate spans >= 5 rings of the centroid grid
// (vplane_10m: 0.5 m voxels, 2 m gate; ndt_10m with 1 m voxels keeps the row loop, which is faster there).
// (the float32 search needs < 80 VGPRs: 6 waves per SIMD.  The float64 centroid search with the row-occupancy
// bitmap wants ~110: bounded to 5 waves it spills 92 bytes per lane; at 4 waves it is the fastest form measured --
// nn time over the trajectory, base / bitmap from ring 2 at 5 waves / the same at 4 waves / bitmap from ring 1 at 4 waves:
// vplane_10m 7105 / 6538 / 6104 / 5978 us, ndt_10m 3777 / 4299 / 3982 / 3903 us)
#ifndef PCR_VOX_WAVES
#define PCR_VOX_WAVES 4
#endif
template <int VOXEL, int HALO, int LOCAL, int MODE, int RB = 0>
__global__ void __launch_bounds__(256, VOXEL == 2 ? PCR_VOX_WAVES : (RB ? 4 : 5)) k_nn_scan(const LinArgs a) {       // RB: see nn_point
    PoseK P;
    PoseQ Q;
    if (!load_pose<false>(a, P)) return;
    if (MODE != PCR_NN_FULL) load_prev(a, Q);
    if (MODE == PCR_NN_LIST) {
        __shared__ uint16_t lst[PCR_LIST_CHUNK];
        __shared__ uint32_t lst_n;
        nn_chunk_loop(a, [&](int64_t first, int64_t end) { nn_chunk_list<VOXEL, HALO>(a, P, Q, lst, &lst_n, first, end); });
    } else {
        // (the device-resident loop: the list set of this iteration was decided by k_gn_update.  Round 6: also under LOCAL == 1 --
        // since round 5's chunk interleave made the block-local hand-out the policy of every mid-size scan, the loop's searches
        // ran as LOCAL == 1 launches and read the FIRST list set at every iteration; select_lists is a no-op for host-driven passes)
        const Geom<float> gsel = (!VOXEL && HALO && LOCAL != 0 && MODE == PCR_NN_FULL) ? select_lists(a) : a.gf;
        auto body = [&](int64_t first, int64_t end) {
            const int64_t i = first + (threadIdx.x & 63);
            if (i < end) nn_point<VOXEL, HALO, MODE == PCR_NN_TRACK, RB>(a, gsel, P, Q, i);
        };
        if (LOCAL == 2) {       // device-resident loop: k_gn_update decided from the size of its step (PoseDev::tile_local)
            if (__builtin_amdgcn_readfirstlane(a.pose->tile_local)) nn_tile_loop<1, 64>(a, body);
            else nn_tile_loop<0, 64>(a, body);
        } else {
            nn_tile_loop<LOCAL, 64>(a, body);
        }
    }
}

#ifdef PCR_DEV
// Plain pass over a POINT target at a far pose: wave-cooperative search with an MFMA distance filter (nn_mfma.h); same
// matches as k_nn_scan<0, HALO, LOCAL, PCR_NN_FULL>, bit for bit
__global__ void __launch_bounds__(256) k_nn_bound(const LinArgs a) {
    PoseK P;
    if (!load_pose<false>(a, P)) return;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * 256) nn_mfma_bound(a, a.gf, P, i);
}
template <int HALO, int LOCAL>
__global__ void __launch_bounds__(256, 4) k_nn_mfma(const LinArgs a) {
    PoseK P;
    if (!load_pose<false>(a, P)) return;
    __shared__ uint32_t cidx_all[4][PCR_MF_MAXC];
    __shared__ uint32_t defer_all[4][64];
    MfWave w;
    w.cidx = cidx_all[threadIdx.x >> 6]; w.defer = defer_all[threadIdx.x >> 6]; w.ndefer = 0;
    auto body = [&](int64_t first, int64_t end) { nn_tile_mfma<HALO, 1>(a, a.gf, P, w, a.pts_last, first, end); };
    nn_tile_loop<LOCAL, 64>(a, body);
    if (w.ndefer > 0) nn_mfma_flush<HALO>(a, a.gf, P, w);
}
#ifdef PCR_MF_STATS
extern "C" __attribute__((visibility("default"))) int pcr_mf_stats_read(unsigned long long out[16], int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mf_stats), 16 * sizeof(unsigned long long)) != hipSuccess) return 1;
    if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_mf_stats), z, sizeof z) != hipSuccess) return 1; }
    return 0;
}
#endif
static void launch_nn_mfma(bool halo, int local, dim3 grid, hipStream_t st, const LinArgs &a) {
    const dim3 block(256);
    {   // the bounds first: a streaming launch over the scan
        int64_t nb = (a.n + 255) / 256;
        if (nb > 256 * 16) nb = 256 * 16;
        hipLaunchKernelGGL(k_nn_bound, dim3((unsigned)(nb > 0 ? nb : 1)), block, 0, st, a);
    }
    if (halo) { if (local == 1) hipLaunchKernelGGL((k_nn_mfma<1, 1>), grid, block, 0, st, a); else hipLaunchKernelGGL((k_nn_mfma<1, 0>), grid, block, 0, st, a); }
    else { if (local == 1) hipLaunchKernelGGL((k_nn_mfma<0, 1>), grid, block, 0, st, a); else hipLaunchKernelGGL((k_nn_mfma<0, 0>), grid, block, 0, st, a); }
}
#endif

// Plain pass over a voxel target that has a float32 filter index (pass_device.h: nn_point_filter)
// (Two-way settling inside this kernel -- nn_point_filter<HALO, 1>: the search also tracks the runner-up and a third bound,
// +8 VALU operations per candidate -- was measured on both 10 M configs in round 4: search 570 -> 640 us and 380 -> 429 us per
// pass, more than the pending points cost anywhere.  The fused small-scan kernel, where one extra search chain doubles the
// longest wave, does use it.)
// Q6 (round 5): the same kernel in front of the float64 coordinates of a float64 POINT target (pass_device.h: nn_filter_core)
template <int HALO, int LOCAL, int Q6 = 0>
__global__ void __launch_bounds__(256, 5) k_nn_filter(const LinArgs a) {
    PoseK P;
    if (!load_pose<false>(a, P)) return;
    auto body = [&](int64_t first, int64_t end) {
        const int64_t i = first + (threadIdx.x & 63);
        if (i < end) nn_point_filter<HALO, 0, Q6>(a, P, i);
    };
    nn_tile_loop<LOCAL, 64>(a, body);
}
#ifdef PCR_DEV
// ... and the float64 search of what it could not certify as a launch of its own (round 3; now only in front of the
// unfused developer reduce kernels); returns at once when no lane of this pass asked
__global__ void __launch_bounds__(256) k_nn_fix(const LinArgs a) {
    PoseK P;
    if (!load_pose<false>(a, P)) return;
    if (__builtin_amdgcn_readfirstlane(*(volatile const uint32_t *)a.pending) != a.stamp) return;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * 256) {
        const uint32_t j = a.nn_j[i];
        if (nn_is_pending(j)) nn_point_fix(a, P, i, j & ~PCR_PENDING_BIT);
    }
}
#endif
static void launch_nn_filter(bool halo, int local, bool separate_fix, bool q6, dim3 grid, hipStream_t st, const LinArgs &a) {
    const dim3 block(256);
#define PCR_NF_CASE(H, L) do { if (q6) hipLaunchKernelGGL((k_nn_filter<H, L, 1>), grid, block, 0, st, a); \
                               else hipLaunchKernelGGL((k_nn_filter<H, L, 0>), grid, block, 0, st, a); } while (0)
    // (local == 2, "decided on the device from the size of the step", is not instantiated here: both tile loops in one
    // kernel around the tracking search spill 736 bytes per lane; the device-resident loop keeps the global counters)
    if (halo) { if (local == 1) PCR_NF_CASE(1, 1); else PCR_NF_CASE(1, 0); }
    else { if (local == 1) PCR_NF_CASE(0, 1); else PCR_NF_CASE(0, 0); }
#undef PCR_NF_CASE
    // the float64 search of the pending points: the prologue of k_reduce_finalize<KIND, 1> (round 4); a launch of its own
    // only in front of the unfused developer reduce kernels
#ifdef PCR_DEV
    if (separate_fix) hipLaunchKernelGGL(k_nn_fix, grid, block, 0, st, a);
#else
    (void)separate_fix;
#endif
}

// host-side choice of the instantiation
// (local: 0 global counters, 1 block-local, 2 decided on the device -- plain searches of the device-resident loop only)
template <int VOXEL, int MODE>
static void launch_nn_scan_mode(bool halo, int local, dim3 grid, hipStream_t st, const LinArgs &a) {
    const dim3 block(256);
    // (RB: plain full searches of a point target that carries row-block boxes, Geom::rbox)
    // (LB: a target with heavy cells, Geom::lbox -- it carries no row-block boxes)
    constexpr bool can_rb = VOXEL == 0 && MODE == PCR_NN_FULL;
    const bool lb = can_rb && a.gf.lbox != nullptr;
    const bool rb = can_rb && !lb && a.gf.rbox != nullptr;
#define PCR_NN_CASE(H, L) do { if (lb) hipLaunchKernelGGL((k_nn_scan<VOXEL, (VOXEL ? 0 : H), L, MODE, can_rb ? 2 : 0>), grid, block, 0, st, a); \
                               else if (rb) hipLaunchKernelGGL((k_nn_scan<VOXEL, (VOXEL ? 0 : H), L, MODE, can_rb ? 1 : 0>), grid, block, 0, st, a); \
                               else hipLaunchKernelGGL((k_nn_scan<VOXEL, (VOXEL ? 0 : H), L, MODE, 0>), grid, block, 0, st, a); } while (0)
    if (MODE == PCR_NN_FULL && local == 2) { if (halo) PCR_NN_CASE(1, 2); else PCR_NN_CASE(0, 2); return; }
    if (halo) { if (local) PCR_NN_CASE(1, 1); else PCR_NN_CASE(1, 0); }
    else { if (local) PCR_NN_CASE(0, 1); else PCR_NN_CASE(0, 0); }
#undef PCR_NN_CASE
}
template <int VOXEL>
static void launch_nn_scan(int mode, bool halo, int local, dim3 grid, hipStream_t st, const LinArgs &a) {
    switch (mode) {
    case PCR_NN_FULL: launch_nn_scan_mode<VOXEL, PCR_NN_FULL>(halo, local, grid, st, a); break;
    case PCR_NN_TRACK: launch_nn_scan_mode<VOXEL, PCR_NN_TRACK>(halo, local != 0, grid, st, a); break;
    default: launch_nn_scan_mode<VOXEL, PCR_NN_LIST>(halo, 0, grid, st, a); break;
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
    double T_old[16];
    for (int i = 0; i < 16; ++i) T_old[i] = T[i];
    const int r = gn_step(A, f.out, f.tol, T);
    // hand-out policy of the next search (see pass_enqueue): block-local once the scan moves by less than local_len
    const double moved = r == 0 ? gn_typical_motion(T_old, T, f.bb_c, f.bb_e) : 0.0;
    p->tile_local = (r == 0 && f.local_len > 0.0 && moved < f.local_len) ? 1 : 0;
    p->halo_deep = (r == 0 && moved >= f.deep_len) ? 1 : 0;
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
    if (threadIdx.x != 0) return;
    if (f.pose->done == PCR_LOOP_RUNNING) {
        double A[6][7];                 // registers: gn_solve6 indexes it with compile-time constants only
        gn_update(f, A);
    } else if (f.pose->done == PCR_LOOP_COMMFAIL && f.host_state) {
        // the exchange in front of this launch gave up waiting for a peer: tell the host, which is watching this word
        *f.host_state = ((unsigned long long)(unsigned)PCR_LOOP_COMMFAIL << 32) | (unsigned)f.pose->iter;
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
    p->tile_local = 0;
    p->halo_deep = 1;          // the first pass knows nothing about its distance to the target: the deeper lists (equal there)
}

// The pending points of a filter pass (1-2 per 1000) inside the range of scan points this thread is about to reduce.
// Searched one by one as they turn up, a wave runs one float64 search chain (~20 dependent round trips) per loop
// iteration in which ANY of its lanes has a pending point -- 3-4 chains in sequence per wave on vplane_10m, as long as the
// separate k_nn_fix launch took (measured: reduce 64 -> 144 us).  So the wave first COLLECTS its pending points in a
// small list in LDS (ballot + prefix count, no atomics) and searches them 64 at a time: one chain per wave in the
// ordinary case, dense waves when everything is pending (duplicated centroids).
__device__ __forceinline__ void fix_pending(const LinArgs &a, const PoseK &P, const TileIter &it) {
    __shared__ uint32_t fix_list[4][128];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t *lst = fix_list[wave];
    const int64_t base0 = it.base - threadIdx.x;               // first point of this block's first tile
    const unsigned long long below = (1ull << lane) - 1ull;
    int cnt = 0;                                                // wave-uniform
    // FIX_UNROLL words of nn_j are requested before the first is looked at: one word per round trip made the walk itself a
    // chain of ~38 dependent loads per wave (10 M points over 1024 blocks), 55-60 us -- as long as the searches it feeds
    constexpr int FIX_UNROLL = 8;
    for (int64_t i0 = base0;; i0 += FIX_UNROLL * it.stride) {
        const bool more = i0 < it.end;                          // block-uniform
        if (more) {
            uint32_t jv[FIX_UNROLL];
#pragma unroll
            for (int u = 0; u < FIX_UNROLL; ++u) {
                const int64_t i = i0 + (int64_t)u * it.stride + threadIdx.x;
                jv[u] = i < it.end ? a.nn_j[i] : PCR_NONE;
            }
#pragma unroll
            for (int u = 0; u < FIX_UNROLL; ++u) {
                const bool pend = nn_is_pending(jv[u]);
                const unsigned long long m = __ballot(pend);
                if (m == 0) continue;
                if (pend) lst[cnt + __popcll(m & below)] = (uint32_t)(i0 + (int64_t)u * it.stride + threadIdx.x - base0);
                cnt += __popcll(m);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (cnt >= 64) {                                // a full wave of pending points: search them now
                    const int64_t ip = base0 + (int64_t)lst[lane];
#ifdef PCR_DEV
                    if (a.flags & (1u << 28)) a.nn_j[ip] &= ~PCR_PENDING_BIT;      // (PCR_FIX_DEBUG=1, timing only: the walk without the searches)
                    else
#endif
                    nn_point_fix(a, P, ip, a.nn_j[ip] & ~PCR_PENDING_BIT);
                    const uint32_t carry = lane + 64 < cnt ? lst[lane + 64] : 0u;
                    __builtin_amdgcn_wave_barrier();
                    lst[lane] = carry;
                    cnt -= 64;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
            }
            continue;
        }
        // the end of the range: what is still listed (fewer than 64)
        if (lane < cnt) {
            const int64_t ip = base0 + (int64_t)lst[lane];
#ifdef PCR_DEV
            if (a.flags & (1u << 28)) a.nn_j[ip] &= ~PCR_PENDING_BIT;
            else
#endif
            nn_point_fix(a, P, ip, a.nn_j[ip] & ~PCR_PENDING_BIT);
        }
        // (the words rewritten above are read back by OTHER lanes of this wave in reduce_stream -- same-wave global
        // store -> load ordering holds on gfx950, but say so to the compiler and the memory model explicitly)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        break;
    }
}

// k_reduce with the fold inside (the shipped reduce kernel)
// FIX (behind k_nn_filter): every thread first walks the scan points it is about to reduce and runs the float64 search
// for those the filter left PCR_PENDING (1-2 per 1000), rewriting nn_j; then it streams as usual.  Round 3 did this in a
// launch of its own (k_nn_fix: 40-65 us -- the latency of one float64 search chain with the rest of the chip idle, plus
// a 40 MB read of nn_j); here the chains of the few waves that have one overlap with the streaming of all the others,
// and the search's registers are dead before the 32 accumulators come alive.  The order of the sums is unchanged.
template <int KIND, int FIX>
__global__ void __launch_bounds__(256, FIX ? 4 : 1) k_reduce_finalize(const LinArgs a, const FinArgs f) {   // (FIX: stay at <= 128 VGPRs)
    PoseK P;
    if (!load_pose<true>(a, P)) return;
    const TileIter it(a);
    if (FIX && KIND != PCR_ICP) {              // (PLANE: the float64 search of a float64 point target, quirk Q6)
        if ((uint32_t)__builtin_amdgcn_readfirstlane((int)*(volatile const uint32_t *)a.pending) == a.stamp)
            fix_pending(a, P, it);
    }
    double acc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = 0.0;
    reduce_stream<KIND>(acc, a, P, it.base, it.end, it.stride);
    (void)ticket_fold_emit(acc, a, f);
    // (the Gauss-Newton step is NOT inlined here: its straight-line float64 code needs 136 VGPRs, which
    // would cap this streaming kernel at 3 blocks per CU; k_gn_update runs it as a 1-wave launch)
}

// Round 6 (the tail of an iteration at B-01 size).  Search AND reduce of a mid-size scan over a point
// target in ONE launch -- PHASE-SPLIT, not interleaved: the fused small-scan kernel (k_linearize_finalize) keeps the 32
// float64 accumulators alive across every search and pays for them with half the occupancy of k_nn_scan (measured slower from
// ~300 k points on); here a block first SEARCHES all the tiles of its block-local hand-out exactly as k_nn_scan<.., LOCAL = 1>
// does (matches to nn_j, 73-80 VGPRs' worth of live state), meets at one barrier, and only then walks the same tile list
// again -- dealt to its four waves STATICALLY, so the order of every sum is fixed whatever the timing of the search -- to
// gather and accumulate, folds and takes its ticket.  The accumulators exist only in the second phase: the kernel's register
// count is the maximum of the two phases, not their sum.  What it saves is the reduce kernel as a LAUNCH: at 1.06 M points
// k_reduce_finalize is a 22-us chain (launch, cold gathers, fold, two ticket levels) of which the stream itself is a third;
// inside the search kernel that work runs in the shadow of the other blocks' searches and only the last block's fold is left.
// The sums differ from the search + reduce pair's in the last bits (another association of the same terms), so a context
// runs ONE of the two forms for a given scan size, and the certified-reuse modes (whose tests compare sums across modes bit
// for bit) keep the pair.
#ifndef PCR_PS_WAVES
#define PCR_PS_WAVES 4      // (5: the reduce phase spills 104 bytes per lane and the pass is slower still, profiles/r06_phase_split_null.txt)
#endif
template <int KIND, int HALO>
__global__ void __launch_bounds__(256, PCR_PS_WAVES) k_scan_reduce(const LinArgs a, const FinArgs f) {
    static_assert(KIND == PCR_ICP || KIND == PCR_PLANE, "point targets only");
    PoseK P;
    if (!load_pose<false>(a, P)) return;
    {
        PoseQ Q;                                    // (tracking searches only: unused)
        const Geom<float> gsel = HALO ? select_lists(a) : a.gf;
        nn_tile_loop<1, 64>(a, [&](int64_t first, int64_t end) {
            const int64_t i = first + (threadIdx.x & 63);
            if (i < end) nn_point<0, HALO, 0, 0>(a, gsel, P, Q, i);
        });
    }
    // the block's matches are in nn_j: workgroup-scope release / acquire around the barrier (one CU, one L1)
    __syncthreads();
    // (the float64 rotation only now: 18 scalar registers the search has no room for)
    if (a.pose == nullptr) {
#pragma unroll
        for (int i = 0; i < 9; ++i) P.R[i] = a.hp.R[i];
    } else {
#pragma unroll
        for (int i = 0; i < 9; ++i) P.R[i] = uniform_f64(a.pose->R[i]);
    }
    // The accumulators are SPLIT over the two halves of the wave: lanes l and l + 32 take the SAME scan point (32 points of a
    // tile per step), the lower lane carries components 0..15 of the 32-vector, the upper lane components 16..31 -- 16 float64
    // accumulators per lane instead of 32 (the whole-vector form needs > 96 VGPRs: 251 spilled at 5 waves / SIMD).  That is
    // exactly the state wave_fold32 is in after its first halving step, so the fold simply starts at the second one.
    double acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0;
    {
        // the tile list of nn_tile_loop<1, 64> (virtual tile xb + k nxb of this XCD, chunk-interleaved), wave w takes
        // k = w, w + 4, ...; both halves' gathers in flight, accumulated in index order
        const int xcd = (int)(blockIdx.x & 7);
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const bool upper = lane >= 32;
        const int64_t xb = blockIdx.x >> 3, nxb = gridDim.x >> 3;
        const int64_t chunk = (int64_t)PCR_TILE_CHUNK * 64;
        const int64_t nchunks = (a.n + chunk - 1) / chunk;
        const int64_t vspan = ((nchunks + 7) / 8) * chunk;
        for (int64_t k = wave;; k += 4) {
            const int64_t vfirst = (xb + k * nxb) * 64;
            if (vfirst >= vspan) break;
            const int64_t first = nn_tile_real<64>(a, xcd, 0, vfirst);
            if (first >= a.n) break;                // (wave-uniform: real positions grow with k)
#pragma unroll 1
            for (int u = 0; u < 2; ++u) {
                const int64_t ipu = first + (lane & 31) + 32 * u;
                const uint32_t ju = ipu < a.n ? a.nn_j[ipu] : PCR_NONE;
                if (ju == PCR_NONE) continue;
                float4 qu, nru = make_float4(0, 0, 0, 0);
                if (KIND == PCR_PLANE) { const float4 *r = reinterpret_cast<const float4 *>(a.pn + ju); qu = r[0]; nru = r[1]; }
                else qu = a.pts[ju];
                const float xf = a.sx[ipu], yf = a.sy[ipu], zf = a.sz[ipu];
                float tx, ty, tz;
                xform(P, xf, yf, zf, tx, ty, tz);
                const float dxf = tx - qu.x, dyf = ty - qu.y, dzf = tz - qu.z;
                if (!gate_f32(a, dxf, dyf, dzf)) continue;
                const double x = xf, y = yf, z = zf, d0 = dxf, d1 = dyf, d2 = dzf;
                if (KIND == PCR_PLANE) {
                    // acc_plane / acc_rank1, component by component (plane_icp.py:49-67)
                    const double n0 = nru.x, n1 = nru.y, n2 = nru.z;
                    const double r = (n0 * d0 + n1 * d1) + n2 * d2;
                    const double ra = P.R[0] * n0 + P.R[3] * n1 + P.R[6] * n2;
                    const double rb = P.R[1] * n0 + P.R[4] * n1 + P.R[7] * n2;
                    const double rc = P.R[2] * n0 + P.R[5] * n1 + P.R[8] * n2;
                    const double J3 = -z * rb + y * rc, J4 = z * ra - x * rc, J5 = -y * ra + x * rb;
                    if (!upper) {
                        acc[0] = fma(n0, n0, acc[0]); acc[1] = fma(n0, n1, acc[1]); acc[2] = fma(n0, n2, acc[2]);
                        acc[3] = fma(n0, J3, acc[3]); acc[4] = fma(n0, J4, acc[4]); acc[5] = fma(n0, J5, acc[5]);
                        acc[6] = fma(n1, n1, acc[6]); acc[7] = fma(n1, n2, acc[7]); acc[8] = fma(n1, J3, acc[8]);
                        acc[9] = fma(n1, J4, acc[9]); acc[10] = fma(n1, J5, acc[10]);
                        acc[11] = fma(n2, n2, acc[11]); acc[12] = fma(n2, J3, acc[12]); acc[13] = fma(n2, J4, acc[13]);
                        acc[14] = fma(n2, J5, acc[14]);
                        acc[15] = fma(J3, J3, acc[15]);
                    } else {
                        acc[0] = fma(J3, J4, acc[0]); acc[1] = fma(J3, J5, acc[1]);
                        acc[2] = fma(J4, J4, acc[2]); acc[3] = fma(J4, J5, acc[3]);
                        acc[4] = fma(J5, J5, acc[4]);
                        acc[5] = fma(n0, r, acc[5]); acc[6] = fma(n1, r, acc[6]); acc[7] = fma(n2, r, acc[7]);
                        acc[8] = fma(J3, r, acc[8]); acc[9] = fma(J4, r, acc[9]); acc[10] = fma(J5, r, acc[10]);
                        acc[11] = fma(r, r, acc[11]);
                        acc[12] += 1.0;
                    }
                } else {
                    // acc_icp's 17 components: 0..15 on the lower lane, e2 (16) on the upper one
                    if (!upper) {
                        double full[32];
#pragma unroll
                        for (int i = 0; i < 16; ++i) full[i] = acc[i];
                        full[16] = 0.0;
                        acc_icp(full, P, a.flags, x, y, z, d0, d1, d2);
#pragma unroll
                        for (int i = 0; i < 16; ++i) acc[i] = full[i];
                    } else {
                        acc[0] += d0 * d0 + d1 * d1 + d2 * d2;
                    }
                }
            }
        }
    }
    (void)ticket_fold_emit<true>(acc, a, f);
}

// GN (device-resident loop on one GPU): the block that emits also takes the Gauss-Newton step.  This kernel runs one
// tile per wave at 129-191 VGPRs anyway, so -- unlike in the streaming reduce kernel -- the step's registers cost no
// occupancy, and the iteration saves the k_gn_update launch (~8 us of a 46 us iteration on a 100 k-point scan).  Every
// other block read the pose before it contributed its ticket, so rewriting it here races with nothing.
template <int KIND, int HALO, int GN, int FILT, int LB = 0>
__global__ void __launch_bounds__(256) k_linearize_finalize(const LinArgs a, const FinArgs f) {
    PoseK P;
    if (!load_pose<true>(a, P)) return;
    double acc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = 0.0;
    linearize_body<KIND, HALO, FILT, LB>(a, P, acc);
    const bool last = ticket_fold_emit(acc, a, f);
    if (GN && last && threadIdx.x == 0 && f.pose->done == PCR_LOOP_RUNNING) {
        double A[6][7];
        gn_update(f, A);
    }
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
template <typename Real, typename PT, bool HALO, bool LB = false>
__global__ void __launch_bounds__(256) k_nn_query(Geom<Real> g, const PT *pts, const uint32_t *cs,
                                                  const float *q, int64_t m, Real bound2, Real rmax,
                                                  Real *dist, int64_t *idx) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    Real best; uint32_t bj, bo;
    nn_search<Real, PT, false, false, HALO, 0, false, PCR_NN_BATCH, false, LB>(g, pts, cs, (Real)q[3 * i], (Real)q[3 * i + 1], (Real)q[3 * i + 2], bound2, best, bj, bo);
    bj = bo;                                       // (only tested against PCR_NONE below)
    Real d = RealTraits<Real>::sqrt_rn(best);
    if (bj != PCR_NONE && rmax < RealTraits<Real>::inf() && !(d < rmax)) bj = PCR_NONE;
    dist[i] = bj == PCR_NONE ? RealTraits<Real>::inf() : d;
    idx[i] = bj == PCR_NONE ? (int64_t)-1 : (int64_t)bo;
}

// KDTree(float64 data).query (kdtree.py:18-21; quirk Q6): the float32 search over the index nominates, the float64 box search
// through the nominee (nn_box_f64: every point whose float64 position can lie inside that ball) decides by (float64
// distance, original index).  The fine seam: no certification shortcut, every query runs both.
template <bool HALO>
__global__ void __launch_bounds__(256) k_nn_query_q6(Geom<float> gf, const PtF *pts, const uint32_t *cs, Geom<double> gq, const PtD *pts64,
                                                     const float *q, int64_t m, float bound2_f, double rmax, double *dist, int64_t *idx) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    const float qx = q[3 * i], qy = q[3 * i + 1], qz = q[3 * i + 2];
    float best; uint32_t bj, bo;
    nn_search<float, PtF, false, false, HALO>(gf, pts, cs, qx, qy, qz, bound2_f, best, bj, bo);
    double bd = __longlong_as_double(0x7ff0000000000000LL);
    if (bo != PCR_NONE) nn_box_f64(gq, pts64, cs, (double)qx, (double)qy, (double)qz, bj, bd, bj, bo);
    const double d = __builtin_sqrt(bd);
    if (bo != PCR_NONE && !(d < rmax)) bo = PCR_NONE;
    dist[i] = bo == PCR_NONE ? __longlong_as_double(0x7ff0000000000000LL) : d;
    idx[i] = bo == PCR_NONE ? (int64_t)-1 : (int64_t)bo;
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
        // 8 + 1 tickets (a 128-byte line each), then the tile counters
        // ... then the word of k_nn_fix (the stamp of the last filter pass that left work for it)
        const size_t ctr_words = 9 * PCR_TICKET_STRIDE + (size_t)PCR_TILE_CTRS * PCR_TILE_STRIDE + 16;
        HIP_TRY(pcr_malloc_retry((void **)&ctx->d_tile_ctr, sizeof(uint32_t) * ctr_words));
        HIP_TRY(hipMemsetAsync(ctx->d_tile_ctr, 0, sizeof(uint32_t) * ctr_words, ctx->stream));
#ifdef PCR_DEV
        {
            int nbm = 0;
            const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nbm, k_nn_mfma<1, 0>, 256, 0);
            ctx->nn_blocks_per_cu[4] = (e == hipSuccess && nbm > 0) ? nbm : 2;
        }
#endif
        {
            int nb = 0;
            const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_nn_scan<0, 1, 0, 0, 1>, 256, 0);
            ctx->nn_blocks_rb = (e == hipSuccess && nb > 0) ? nb : 4;
            const hipError_t e2 = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_nn_scan<0, 1, 0, 0, 2>, 256, 0);
            ctx->nn_blocks_lb = (e2 == hipSuccess && nb > 0) ? nb : 4;
            const hipError_t e3 = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_scan_reduce<PCR_PLANE, 1>, 256, 0);
            ctx->nn_blocks_ps = (e3 == hipSuccess && nb > 0) ? nb : 4;
        }
        for (int v = 0; v < 4; ++v) {
            int nb = 0;
            if (v == 2) {
#ifdef PCR_DEV
                ctx->nn_blocks_per_cu[v] = pcr_dev_coop_blocks_per_cu();
#endif
                continue;
            }
            hipError_t e = v == 0 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_nn_scan<0, 1, 0, 0>, 256, 0)
                         : v == 1 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_nn_scan<1, 0, 0, 0>, 256, 0)
                                  : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_nn_filter<1, 0>, 256, 0);
            ctx->nn_blocks_per_cu[v] = (e == hipSuccess && nb > 0) ? nb : 4;
        }
    }
    return PCR_OK;
}

int choose_blocks(const pcr_context *ctx, int64_t n) {
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
    bool gn_inline;      // device-resident loop, fused kernel, one GPU: the Gauss-Newton step runs inside k_linearize_finalize
    double motion;       // typical displacement of the scan since the previous pass over it (host-driven passes; -1 unknown)
    int nn_mode;         // PCR_NN_FULL / TRACK / LIST
    bool reuse_ready;    // the scan has the buffers of the certified-reuse path
    bool q6;             // PlaneICP pass over a float64 point target: float64 search (quirk Q6, pcr_target::pts64)
};

bool pcr_pass_is_fused(const pcr_context *ctx, const pcr_scan *s) {
    if (ctx->variant == 2) return s->n <= (int64_t)ctx->num_cu * 1024;   // measured crossover: 262 k - 350 k points on 256 CUs
    return ctx->variant == 0;
}

static void set_filter_bound(LinArgs &a, double bound);
// (bits 27-29 of LinArgs::flags are internal -- the gate switch of quirk Q6, two developer timing switches; a
// caller-supplied word must not reach them)
#define PCR_PUBLIC_FLAGS (PCR_FLAG_ICP_RR_QUIRK | PCR_FLAG_NO_SCAN_SORT | PCR_FLAG_LOCAL_ONLY | PCR_FLAG_HOST_LOOP | PCR_FLAG_DEVICE_LOOP)
static pcr_status pass_setup(Pass *ps, pcr_target *t, pcr_scan *s, int kind, double max_dist, unsigned flags) {
    pcr_context *ctx = t->ctx;
    flags &= PCR_PUBLIC_FLAGS;
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
    // quirk Q6: a PlaneICP pass over a float64 point target searches in float64 -- always search + reduce, always a full search
    const bool q6 = kind == PCR_PLANE && !t->is_voxel && t->pts64 != nullptr;
    const bool one_kernel = pcr_pass_is_fused(ctx, s) && !q6;
    if (!one_kernel && !s->nn_j) {
        HIP_TRY(pcr_scan_alloc(s, (void **)&s->nn_j, sizeof(uint32_t) * (size_t)(s->n > 0 ? s->n : 1)));
#ifdef PCR_EXP_SEED
        HIP_TRY(hipMemsetAsync(s->nn_j, 0xff, sizeof(uint32_t) * (size_t)(s->n > 0 ? s->n : 1), ctx->stream));
#endif
        s->nn_serial = 0;
    }
    const int nblocks_split = [&] {
        int nb = choose_blocks(ctx, s->n);
        if (nb > ctx->num_cu * 4) nb = ctx->num_cu * 4;
        nb &= ~7;
        return nb < 8 ? 8 : nb;
    }();
    ps->reuse_ready = false;
    const bool nn_plain = ctx->nn_mode == 0 || ctx->nn_mode == 4;     // (4, developer build: plain searches run the MFMA-filtered kernel)
    if (!one_kernel && !q6 && ctx->reuse != 0 && (nn_plain || ctx->nn_mode == 3) && s->n > 0) {
        if (!s->lb2 || !s->umask) {
            const size_t words = (((size_t)s->n + 63) / 64 + 31) & ~(size_t)15;     // whole 16-word chunks + slack
            if (!s->lb2) HIP_TRY(pcr_scan_alloc(s, (void **)&s->lb2, sizeof(float) * (size_t)s->n));
            HIP_TRY(pcr_scan_alloc(s, (void **)&s->umask, sizeof(unsigned long long) * words));
            HIP_TRY(hipMemsetAsync(s->umask, 0, sizeof(unsigned long long) * words, ctx->stream));
            s->track_valid = false;
        }
        if (s->ucnt_cap < nblocks_split) {
            pcr_scan_free(s, s->ucnt);
            s->ucnt = nullptr; s->ucnt_cap = 0;
            HIP_TRY(pcr_scan_alloc(s, (void **)&s->ucnt, sizeof(uint32_t) * (size_t)nblocks_split));
            s->ucnt_cap = nblocks_split;
        }
        ps->reuse_ready = true;
    }
    // the MFMA-filtered search (developer build) keeps its per-point bounds in lb2 (dead between passes unless the previous one
    // was a tracking pass, and a full search invalidates those anyway)
    if (!one_kernel && !q6 && !t->is_voxel && ctx->nn_mode == 4 && !s->lb2 && s->n > 0) {
        HIP_TRY(pcr_scan_alloc(s, (void **)&s->lb2, sizeof(float) * (size_t)s->n));
        s->track_valid = false;
    }
    ps->ctx = ctx; ps->t = t; ps->s = s; ps->kind = kind; ps->one_kernel = one_kernel; ps->q6 = q6;
    LinArgs &a = ps->a;
    memset(&a, 0, sizeof a);
    a.sx = s->x; a.sy = s->y; a.sz = s->z; a.n = s->n;
    a.gf = t->gf; a.pts = t->pts; a.pn = t->pn;
    a.pts_last = (uint32_t)(t->is_voxel ? 0 : t->n + PCR_PTS_PAD - 1);
    a.gd = t->gd; a.means = t->means; a.vnorm = t->vnorm; a.vicov = t->vicov;
    a.cell_start = t->cell_start;
    // The filter index is built by the first pass that gets its cost back: a search + reduce pass at once; the fused
    // small-scan kernel (which gains ~10 us per pass from it against ~0.3 ms of build) once the target has served
    // PCR_FILTER_AFTER fused passes -- i.e. not during the one align() of the reference's benchmark protocol
    // (set_target + align, benchmark/speed_test_comparison.py:36-55), but from the second align on for a map that stays
    const bool want_filter = t->is_voxel && ctx->vox_filter && ctx->nn_mode != 3 && (!one_kernel || ctx->fuse_finalize);
    if (want_filter && one_kernel && !t->filter_tried) ++t->fused_passes;
    if (want_filter && !t->filter_tried && (!one_kernel || t->fused_passes > ctx->filter_after)) {
        t->filter_tried = true;
        // a failed build (out of memory, say) must not fail the pass: the float64 search needs no filter
        if (pcr_build_centroid_filter(ctx, t) != PCR_OK || (t->filter && !(t->filter_band > 0))) {
            fprintf(stderr, "[pcr] the float32 filter index of a voxel target could not be built (%s); its centroid searches stay in float64\n",
                    pcr_last_error());
            (void)hipGetLastError();
            pcr_target_release(t->filter);
            t->filter = nullptr; t->filter_band = 0;
        }
    }
    if (t->is_voxel && t->filter && ctx->vox_filter && ctx->nn_mode != 3 && (!one_kernel || ctx->fuse_finalize)) {
        a.gf = t->filter->gf; a.pts = t->filter->pts; a.cs_f = t->filter->cell_start;
        a.band_f = (float)(t->filter_band * 1.000001);
    }
    if (q6) {
        // the filter is the target's own float32 index (a.gf / a.pts / cell_start as they are); the float64 side reads the
        // float64 coordinates in the same order through the voxel-target fields of the kernels
        a.gd = t->gq; a.means = t->pts64; a.cs_f = t->cell_start;
        a.band_f = (float)(t->band64 * 1.000001);
        flags |= PCR_IFLAG_NOGATE;
    }
    // the deeper set of extended lists of a point target: built once the target has served PCR_HALO2_AFTER search + reduce
    // passes (a failed build is not an error: the pass runs on the first set)
    if (!t->is_voxel && !one_kernel && !q6 && t->cs_h && nn_plain) {
        if (!t->deep_tried && ++t->split_passes > PCR_HALO2_AFTER) {
            t->deep_tried = true;
            if (pcr_build_deep_lists(ctx, t) != PCR_OK) (void)hipGetLastError();
        }
        if (t->cs_h2) { a.cs_h2 = t->cs_h2; a.pts_h2 = t->pts_h2; a.j_h2 = t->j_h2; a.halo2_f = t->halo2; a.lbox_h2 = t->lbox_h2; a.gbox_h2 = t->gbox_h2; }
    }
    a.md_f = (float)max_dist; a.md_d = max_dist;
    const double bound = max_dist * (1.0 + 1e-6);
    a.bound2_f = (float)(bound * bound); a.bound2_d = bound * bound;
    set_filter_bound(a, bound);
    a.flags = flags;           // (public bits were masked by the callers below: PCR_PUBLIC_FLAGS; internal bits are OR-ed in above)
    a.nblocks = choose_blocks(ctx, s->n);
    a.partials = ctx->d_partials;
    a.nn_j = s->nn_j; a.tile_ctr = ctx->d_tile_ctr + 9 * PCR_TICKET_STRIDE;
    a.pending = ctx->d_tile_ctr + 9 * PCR_TICKET_STRIDE + (size_t)PCR_TILE_CTRS * PCR_TILE_STRIDE;
    a.lb2 = s->lb2; a.umask = s->umask; a.ucnt = s->ucnt;
    a.mu_f = (float)(ctx->reuse_mu * (t->is_voxel ? t->gd.h : (double)t->gf.h));
    if (!one_kernel && a.nblocks > ctx->num_cu * 4) a.nblocks = ctx->num_cu * 4;   // k_reduce streams: 4 blocks/CU
    // TileIter and the ticket counts of k_reduce_finalize need a multiple of 8 blocks
    a.nblocks &= ~7;
    if (a.nblocks < 8) a.nblocks = 8;
#ifdef PCR_DEV
    ps->fused_fin = ctx->fuse_finalize;
#else
    ps->fused_fin = true;                       // (the unfused folds live in kernels_dev.hip: developer build only)
#endif
    ps->nn_mode = PCR_NN_FULL;
    ps->motion = -1.0;
    ps->gn_inline = false;
    FinArgs &f = ps->f;
    memset(&f, 0, sizeof f);
    for (int i = 0; i < 3; ++i) { f.bb_c[i] = s->bb_c[i]; f.bb_e[i] = s->bb_e[i]; }
    f.local_len = ctx->local_frac * (t->is_voxel ? t->gd.h : (double)t->gf.h);
    f.deep_len = PCR_HALO2_MOVE * (t->is_voxel ? t->gd.h : (double)t->gf.h);
    f.ucnt = s->ucnt; f.n_ucnt = a.nblocks;
    f.partials = ctx->d_partials; f.tile_ctr = ctx->d_tile_ctr + 9 * PCR_TICKET_STRIDE; f.tickets = ctx->d_tile_ctr; f.nblocks = a.nblocks; f.kind = kind; f.out = ctx->d_out;
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

// bound and margin of the float32 filter search (nn_point_filter) that go with a float64 search bound
static void set_filter_bound(LinArgs &a, double bound) {
    if (a.band_f <= 0.f) return;
    gn_filter_bounds((double)a.band_f, bound, &a.bound2_ff, &a.mu_ff);
}

// What the search of a pass does (gn_math.h: gn_choose_nn_mode) and the search bound that goes with it: a tracking
// search looks 5 % beyond the gate, so that a point with nothing in reach can be certified "still nothing" later.
static void pass_set_mode(Pass *ps, int mode) {
    ps->nn_mode = mode;
    ps->f.nn_mode = mode;
    const double md = ps->a.md_d;
    const double bound = mode == PCR_NN_FULL ? md * (1.0 + 1e-6) : md * 1.05;
    ps->a.bound2_f = (float)(bound * bound); ps->a.bound2_d = bound * bound;
    set_filter_bound(ps->a, bound);
}

static int host_choose_mode(const Pass *ps, const double T[16], double *motion_out) {
    const pcr_context *ctx = ps->ctx;
    const pcr_scan *s = ps->s;
    const pcr_target *t = ps->t;
    *motion_out = -1.0;
    const int have_prev = s->pose_valid && s->nn_serial == t->serial && s->nn_serial != 0;
    if (!have_prev) return PCR_NN_FULL;
    const double h = t->is_voxel ? t->gd.h : (double)t->gf.h;
    const double m = gn_typical_motion(s->prev_T, T, s->bb_c, s->bb_e);
    *motion_out = m;
    if (!ps->reuse_ready) return PCR_NN_FULL;
    return gn_choose_nn_mode(ctx->reuse, have_prev, s->track_valid ? 1 : 0, m, s->last_motion, ctx->reuse_tau * h);
}

template <int KIND>
static void launch_reduce_kind(const Pass *ps, bool fused, bool fix, dim3 grid) {
    if constexpr (KIND != PCR_ICP) {
        if (fused && fix) {
            hipLaunchKernelGGL((k_reduce_finalize<KIND, 1>), grid, dim3(256), 0, ps->ctx->stream, ps->a, ps->f);
            return;
        }
    }
#ifdef PCR_DEV
    if (!fused) { pcr_dev_launch_reduce(KIND, grid, ps->ctx->stream, ps->a); return; }
#endif
    hipLaunchKernelGGL((k_reduce_finalize<KIND, 0>), grid, dim3(256), 0, ps->ctx->stream, ps->a, ps->f);
}

// enqueue the kernels of one pass on the context's stream (no waiting)
static pcr_status pass_enqueue(Pass *ps) {
    pcr_context *ctx = ps->ctx;
#ifdef PCR_DEV
    {   // developer timing experiments on the pending-point prologue (results are WRONG when set)
        static const int dbg = getenv("PCR_FIX_DEBUG") ? atoi(getenv("PCR_FIX_DEBUG")) : 0;
        if (dbg) ps->a.flags |= (unsigned)(dbg & 3) << 28;
    }
#endif
    const LinArgs &a = ps->a;
    const dim3 grid(a.nblocks), block(256);
    ProfEvent ev;
    if (ctx->prof_on) ctx->prof_this_pass = (ctx->prof_pass++ % (uint64_t)ctx->prof_period) == 0;
    if (ps->one_kernel) {
        pcr_prof_begin(ctx, PCR_K_LINEARIZE, &ev);
        RoctxRange range("pcr:linearize");
        // point targets: HALO = the target has the extended lists; voxel targets: FILT = float32 filter search over the
        // rounded centroids (HALO then refers to the FILTER index)
        const bool filt = ps->t->is_voxel && a.band_f > 0.f && ps->fused_fin;
        const bool halo = ps->t->is_voxel ? (filt && ps->t->filter->cs_h != nullptr) : ps->t->cs_h != nullptr;
        const bool lbf = !ps->t->is_voxel && a.gf.lbox != nullptr;      // heavy point target: ranges through their leaf / group boxes
#define PCR_LIN_LAUNCH(K, H, G, F) do { if constexpr ((K) == PCR_ICP || (K) == PCR_PLANE) { \
            if (lbf) { hipLaunchKernelGGL((k_linearize_finalize<K, H, G, F, 1>), grid, block, 0, ctx->stream, a, ps->f); break; } } \
        hipLaunchKernelGGL((k_linearize_finalize<K, H, G, F, 0>), grid, block, 0, ctx->stream, a, ps->f); } while (0)
#define PCR_LIN_CASE_F(K, F)                                                      \
        if (ps->gn_inline) { if (halo) PCR_LIN_LAUNCH(K, 1, 1, F); else PCR_LIN_LAUNCH(K, 0, 1, F); }  \
        else { if (halo) PCR_LIN_LAUNCH(K, 1, 0, F); else PCR_LIN_LAUNCH(K, 0, 0, F); }
#define PCR_LIN_CASE(K) PCR_LIN_CASE_F(K, 0)
#define PCR_LIN_CASE_V(K) if (filt) { PCR_LIN_CASE_F(K, 1) } else { PCR_LIN_CASE_F(K, 0) }
#ifdef PCR_DEV
        if (!ps->fused_fin) {
            pcr_dev_launch_linearize(ps->kind, halo, grid, ctx->stream, a);
        } else
#endif
        switch (ps->kind) {
        case PCR_ICP: PCR_LIN_CASE(PCR_ICP) break;
        case PCR_PLANE: PCR_LIN_CASE(PCR_PLANE) break;
        case PCR_VPLANE: PCR_LIN_CASE_V(PCR_VPLANE) break;
        default: PCR_LIN_CASE_V(PCR_NDT) break;
        }
#undef PCR_LIN_CASE
#undef PCR_LIN_CASE_V
#undef PCR_LIN_CASE_F
#undef PCR_LIN_LAUNCH
        pcr_prof_end(ctx, &ev);
    } else {
        const bool vox = ps->t->is_voxel != 0;
        const int mode = ps->nn_mode;
        const bool filter = (vox && mode == PCR_NN_FULL && a.band_f > 0.f) || ps->q6;
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
            int64_t nb = (int64_t)ctx->num_cu * ctx->nn_blocks_per_cu[filter ? 3 : vox ? 1 : (ctx->nn_mode == 2 && !ps->q6 ? 2 : 0)];
            if (!filter && !vox && mode == PCR_NN_FULL && ctx->nn_mode != 2) {
                if (a.gf.lbox != nullptr) nb = (int64_t)ctx->num_cu * ctx->nn_blocks_lb;
                else if (a.gf.rbox != nullptr) nb = (int64_t)ctx->num_cu * ctx->nn_blocks_rb;
            }
            // tiles of the hand-out: 64 points per wave, or the 1024-point chunks of a LIST pass (one block per chunk)
            const int64_t tiles = mode == PCR_NN_LIST ? (a.n + PCR_LIST_CHUNK - 1) / PCR_LIST_CHUNK : (a.n + 63) / 64;
            const int64_t need = mode == PCR_NN_LIST ? tiles : (tiles + 3) / 4;
            if (nb > need) nb = need;
            nb = (nb + 7) & ~(int64_t)7;
            if (nb < 8) nb = 8;
            const dim3 nn_grid((unsigned)nb);
            // hand-out policy: at most ~1.5 tiles per launched wave -> block-local (nn_tile_loop)
            // ... or the scan moved little since the previous pass: the far poses are where the cost of a tile varies 10x
            // and the global counters pay (1.06 M points, per pose: 235 / 185 / 108 / 55 / 52 us with the counters,
            // 221 / 203 / 111 / 51 / 40 block-local); the device-resident loop decides in k_gn_update (2)
            int local = tiles * 2 <= nb * 4 * 3 ? 1 : 0;
            // (only while a wave gets a handful of tiles: with 38 tiles per wave -- the 12.5 M-point shard -- the static
            // deal loses whatever the pose: plane_100m 2.93 vs 2.82 ms per pass, vplane_10m 1.075 vs 1.06)
            // Round 5: with the chunk interleave (an XCD's tiles spread over the whole scan) the block-local deal is balanced at the
            // far poses too and wins at EVERY pose of such scans (1.06 M points, search per trajectory: plane_b01 569 -> 531 us, icp_b01
            // 2316 -> 2282, resampled 928 -> 874; forced global counters 590 / 2541 / 966) -- but not with tens of tiles per wave
            // (vplane_10m 3492 -> 5174 us, plane_100m +12 %): profiles/r05_handout_policy.txt
            if (!local && mode != PCR_NN_LIST && tiles <= nb * 4 * 8) {
                if (PCR_TILE_INTERLEAVE) local = 1;
                else if (a.pose != nullptr) local = mode == PCR_NN_FULL ? 2 : 0;
                else if (ps->motion >= 0.0 && ps->motion < ps->f.local_len) local = 1;
            }
            if (ctx->tile_local >= 0) local = ctx->tile_local;
            ps->a.sched_local = local;
            // Round 6: plain full search of a point target with the block-local hand-out -> search + reduce in ONE phase-split
            // launch (k_scan_reduce).  Not for the certified-reuse modes (their tests compare sums across modes bit for bit and
            // the two forms associate the same terms differently), heavy targets (leaf boxes: 111 VGPRs) or developer searches.
            if (ctx->phase_split && !vox && !ps->q6 && mode == PCR_NN_FULL && ctx->nn_mode == 0 && ps->fused_fin && ctx->reuse == 0 &&
                local == 1 && a.gf.lbox == nullptr && a.gf.rbox == nullptr && a.n > 0) {
                int64_t nbs = (int64_t)ctx->num_cu * ctx->nn_blocks_ps;
                {   // (developer: more than one resident generation of blocks, PCR_PS_GRID_MULT)
                    static const double mult = getenv("PCR_PS_GRID_MULT") ? atof(getenv("PCR_PS_GRID_MULT")) : 1.0;
                    if (mult > 0.0) nbs = (int64_t)((double)nbs * mult);
                }
                if (nbs > need) nbs = need;
                nbs = (nbs + 7) & ~(int64_t)7;
                if (nbs < 8) nbs = 8;
                if (nbs <= ctx->max_blocks && tiles <= nbs * 4 * 8) {
                    // host-driven pass: the list set by how far the scan moved since the previous pass (unknown: the deeper one)
                    if (a.pose == nullptr && a.halo2_f > 0.f && !(ps->motion >= 0.0 && ps->motion < ps->f.deep_len)) {
                        ps->a.gf.halo = a.halo2_f; ps->a.gf.cs_h = a.cs_h2; ps->a.gf.pts_h = a.pts_h2; ps->a.gf.j_h = a.j_h2;
                        ps->a.gf.lbox_h = a.lbox_h2; ps->a.gf.gbox_h = a.gbox_h2;
                    }
                    FinArgs f2 = ps->f;
                    f2.nblocks = (int)nbs;                      // tickets and group rows follow THIS grid
                    const dim3 sgrid((unsigned)nbs);
                    if (ev.kernel >= 0) ev.kernel = PCR_K_LINEARIZE;       // (the event opened above: nothing launched under it yet)
                    const bool halo = ps->t->cs_h != nullptr;
                    if (ps->kind == PCR_ICP) {
                        if (halo) hipLaunchKernelGGL((k_scan_reduce<PCR_ICP, 1>), sgrid, block, 0, ctx->stream, a, f2);
                        else hipLaunchKernelGGL((k_scan_reduce<PCR_ICP, 0>), sgrid, block, 0, ctx->stream, a, f2);
                    } else {
                        if (halo) hipLaunchKernelGGL((k_scan_reduce<PCR_PLANE, 1>), sgrid, block, 0, ctx->stream, a, f2);
                        else hipLaunchKernelGGL((k_scan_reduce<PCR_PLANE, 0>), sgrid, block, 0, ctx->stream, a, f2);
                    }
                    pcr_prof_end(ctx, &ev);
                    ps->s->nn_serial = ps->t->serial;
                    HIP_TRY(hipGetLastError());
                    return PCR_OK;
                }
            }
#ifdef PCR_DEV
            // (developer build, nn_mode 4: the MFMA-filtered search on every plain full search of a point target)
            const bool mfma = !vox && !ps->q6 && mode == PCR_NN_FULL && ps->t->n > 0 && ctx->nn_mode == 4;
            if (!vox && !ps->q6 && ctx->nn_mode == 2) {
                pcr_dev_launch_coop(nn_grid, ctx->stream, a);
            } else if (mfma) {
                int64_t nbm = (int64_t)ctx->num_cu * ctx->nn_blocks_per_cu[4];
                if (nbm > need) nbm = need;
                nbm = (nbm + 7) & ~(int64_t)7;
                if (nbm < 8) nbm = 8;
                launch_nn_mfma(ps->t->cs_h != nullptr, ps->a.sched_local == 1 ? 1 : 0, dim3((unsigned)nbm), ctx->stream, a);
            } else
#endif
            if (!vox && !ps->q6) {
                // host-driven pass: the list set by how far the scan moved since the previous pass (unknown: the deeper one)
                if (a.pose == nullptr && a.halo2_f > 0.f && mode == PCR_NN_FULL && !(ps->motion >= 0.0 && ps->motion < ps->f.deep_len)) {
                    ps->a.gf.halo = a.halo2_f; ps->a.gf.cs_h = a.cs_h2; ps->a.gf.pts_h = a.pts_h2; ps->a.gf.j_h = a.j_h2;
                    ps->a.gf.lbox_h = a.lbox_h2; ps->a.gf.gbox_h = a.gbox_h2;
                }
                launch_nn_scan<0>(mode, ps->t->cs_h != nullptr, ps->a.sched_local, nn_grid, ctx->stream, a);
            } else if (filter) {
                if (++ctx->filter_stamp == 0) {                    // (wrapped after 2^32 passes: start over)
                    HIP_TRY(hipMemsetAsync(a.pending, 0, 4, ctx->stream));
                    ctx->filter_stamp = 1;
                }
                ps->a.stamp = ctx->filter_stamp;
                const bool fhalo = ps->q6 ? ps->t->cs_h != nullptr : ps->t->filter->cs_h != nullptr;
                launch_nn_filter(fhalo, ps->a.sched_local == 1 ? 1 : 0, !ps->fused_fin, ps->q6, nn_grid, ctx->stream, a);
            } else if (ps->t->gd.rowocc != nullptr && (ctx->vox_occ >= 0 ? ctx->vox_occ != 0 : a.md_d / ps->t->gd.h + 2.0 >= 5.0)) {
                launch_nn_scan<2>(mode, false, ps->a.sched_local, nn_grid, ctx->stream, a);
            } else {
                launch_nn_scan<1>(mode, false, ps->a.sched_local, nn_grid, ctx->stream, a);
            }
            ps->s->nn_serial = ps->t->serial;      // nn_j now holds matches against this target
        }
        pcr_prof_end(ctx, &ev);
        pcr_prof_begin(ctx, PCR_K_REDUCE, &ev);
        RoctxRange range("pcr:reduce");
        switch (ps->kind) {
        case PCR_ICP: launch_reduce_kind<PCR_ICP>(ps, ps->fused_fin, false, grid); break;
        case PCR_PLANE: launch_reduce_kind<PCR_PLANE>(ps, ps->fused_fin, ps->q6, grid); break;
        case PCR_VPLANE: launch_reduce_kind<PCR_VPLANE>(ps, ps->fused_fin, filter, grid); break;
        default: launch_reduce_kind<PCR_NDT>(ps, ps->fused_fin, filter, grid); break;
        }
        pcr_prof_end(ctx, &ev);
    }
    HIP_TRY(hipGetLastError());
    if (!ps->fused_fin) {
#ifdef PCR_DEV
        pcr_prof_begin(ctx, PCR_K_FINALIZE, &ev);
        pcr_dev_launch_finalize(ctx->stream, ps->f);
        pcr_prof_end(ctx, &ev);
#endif
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
// root-caused in round 3 (tools/stall_study.py, PCR_STALL_DEBUG=1; profiles/archive/r03_stall_root_cause.txt): the calling
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
    ps.motion = motion;
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
        if (use_comm && pcr_comm_failed_now(ctx)) {
            pcr_set_error("peer-to-peer exchange: a peer did not arrive (its sums were not reduced)");
            return PCR_ERR_COMM;
        }
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
    if (use_comm && pcr_comm_failed_now(ctx)) {
        pcr_set_error("peer-to-peer exchange: a peer did not arrive (its sums were not reduced)");
        return PCR_ERR_COMM;
    }
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
    ps.gn_inline = ps.one_kernel && ps.fused_fin && !use_comm;
    auto enqueue_iteration = [&]() -> pcr_status {
        retire_completed(ctx);
        PCR_TRY(pass_enqueue(&ps));
        if (use_comm) {
            ProfEvent ev;
            pcr_prof_begin(ctx, PCR_K_ALLREDUCE, &ev);
            pcr_status cs = pcr_comm_allreduce29(ctx, ctx->d_out, ctx->d_pose);
            pcr_prof_end(ctx, &ev);
            if (cs != PCR_OK) return cs;
        }
        if (!ps.gn_inline) hipLaunchKernelGGL(k_gn_update, dim3(1), dim3(64), 0, ctx->stream, f);
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
    if (use_comm && loop_done() == PCR_LOOP_COMMFAIL) {
        //  this rank's exchange timed out: no top-up -- the peers are gone or late, and whatever they do with
        // their own timeouts, this call must not report a pose
        (void)hipStreamSynchronize(ctx->stream);
        pcr_set_error("peer-to-peer exchange: a peer did not arrive after %d iteration(s)", passes_done());
        return PCR_ERR_COMM;
    }
    if (use_comm) {
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        const int it_end = passes_done();
        const int target = it_end + AHEAD < max_iter ? it_end + AHEAD : max_iter;
        while (enq < target) PCR_TRY(enqueue_iteration());
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    const int done = loop_done(), it = passes_done();
    if (!t->is_voxel && !ps.one_kernel) t->split_passes += it > 1 ? it - 1 : 0;     // (pass_setup counted one)
    for (int i = 0; i < 16; ++i) T_out[i] = ctx->h_out[40 + i];
    if (iterations) *iterations = it;
    if (trace_or_null && it > 0) {
        HIP_TRY(hipMemcpyAsync(trace_or_null, ctx->d_trace, sizeof(double) * 45 * (size_t)it, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    if (use_comm && pcr_comm_failed_now(ctx)) {
        pcr_set_error("peer-to-peer exchange: a peer did not arrive");
        return PCR_ERR_COMM;
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
        if (t->gf.lbox && t->cs_h)
            hipLaunchKernelGGL((k_nn_query<float, PtF, true, true>), grid, block, 0, ctx->stream, t->gf, t->pts, t->cell_start, d_q, m,
                               bound2, bounded ? (float)r_max : inf, (float *)d_dist, d_idx);
        else if (t->gf.lbox)
            hipLaunchKernelGGL((k_nn_query<float, PtF, false, true>), grid, block, 0, ctx->stream, t->gf, t->pts, t->cell_start, d_q, m,
                               bound2, bounded ? (float)r_max : inf, (float *)d_dist, d_idx);
        else if (t->cs_h)
            hipLaunchKernelGGL((k_nn_query<float, PtF, true>), grid, block, 0, ctx->stream, t->gf, t->pts, t->cell_start, d_q, m,
                               bound2, bounded ? (float)r_max : inf, (float *)d_dist, d_idx);
        else
            hipLaunchKernelGGL((k_nn_query<float, PtF, false>), grid, block, 0, ctx->stream, t->gf, t->pts, t->cell_start, d_q, m,
                               bound2, bounded ? (float)r_max : inf, (float *)d_dist, d_idx);
    } else if (!t->is_voxel) {
        PCR_REQUIRE(t->pts64 != nullptr, "pcr_nn_query_f64 needs a voxel target or a point target with float64 coordinates (pcr_target_points_set_f64)");
        // the float32 search must reach every point whose FLOAT64 position lies within r_max: r_max + band, a little more
        const double b = (r_max + t->band64) * 1.00002;
        const float bound2 = bounded ? (float)(b * b * 1.000001) : __builtin_inff();
        const double rmax = bounded ? r_max : __builtin_inf();
        if (t->cs_h)
            hipLaunchKernelGGL((k_nn_query_q6<true>), grid, block, 0, ctx->stream, t->gf, t->pts, t->cell_start, t->gq, t->pts64, d_q, m,
                               bound2, rmax, (double *)d_dist, d_idx);
        else
            hipLaunchKernelGGL((k_nn_query_q6<false>), grid, block, 0, ctx->stream, t->gf, t->pts, t->cell_start, t->gq, t->pts64, d_q, m,
                               bound2, rmax, (double *)d_dist, d_idx);
    } else {
        const double inf = __builtin_inf();
        const double b = r_max * (1.0 + 1e-6);
        hipLaunchKernelGGL((k_nn_query<double, PtD, false>), grid, block, 0, ctx->stream, t->gd, t->means, t->cell_start, d_q, m,
                           bounded ? b * b : inf, bounded ? r_max : inf, (double *)d_dist, d_idx);
    }
    pcr_prof_end(ctx, &ev);
    HIP_TRY(hipGetLastError());
    return PCR_OK;
}

