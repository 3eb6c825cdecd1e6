// Developer / A-B kernels: NOT on the shipped path.  Kept selectable through
// pcr_set_variant / pcr_set_fuse_finalize / pcr_set_nn_mode so that every alternative the design was measured
// against stays runnable and parity-tested: the unfused folds (k_linearize / k_reduce + k_finalize), the
// wave-cooperative LDS-staged search of the north star (k_nn_coop), and the work counters of the search.
#include "pass_device.h"

template <int KIND, int HALO>
__global__ void __launch_bounds__(256) k_linearize(const LinArgs a) {
    PoseK P;
    if (!load_pose<true>(a, P)) return;
    double acc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = 0.0;
    linearize_body<KIND, HALO, 0>(a, P, acc);
    block_store_partials(acc, a.partials);
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
template <int HALO, bool LB = false>
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
        if (live) kstart = nn_ring0<float, PtF, true, HALO != 0, 0, PCR_NN_BATCH, LB>(a.gf, a.pts, a.cell_start, c, tx, ty, tz, best, bj, bo, &st);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        const unsigned long long t2 = __builtin_readcyclecounter();
        if (live) nn_rings<float, PtF, true, 0, PCR_NN_BATCH, LB>(a.gf, a.pts, a.cell_start, c, kstart, tx, ty, tz, best, bj, bo, &st);
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

// ---- host-side launchers (declared in pass_device.h) ------------------------------------------------
void pcr_dev_launch_linearize(int kind, bool halo, dim3 grid, hipStream_t st, const LinArgs &a) {
    const dim3 block(256);
#define PCR_LIN_CASE(K)                                                                       \
    if (halo) hipLaunchKernelGGL((k_linearize<K, 1>), grid, block, 0, st, a);                 \
    else hipLaunchKernelGGL((k_linearize<K, 0>), grid, block, 0, st, a);
    switch (kind) {
    case PCR_ICP: PCR_LIN_CASE(PCR_ICP) break;
    case PCR_PLANE: PCR_LIN_CASE(PCR_PLANE) break;
    case PCR_VPLANE: PCR_LIN_CASE(PCR_VPLANE) break;
    default: PCR_LIN_CASE(PCR_NDT) break;
    }
#undef PCR_LIN_CASE
}

void pcr_dev_launch_reduce(int kind, dim3 grid, hipStream_t st, const LinArgs &a) {
    const dim3 block(256);
    switch (kind) {
    case PCR_ICP: hipLaunchKernelGGL(k_reduce<PCR_ICP>, grid, block, 0, st, a); break;
    case PCR_PLANE: hipLaunchKernelGGL(k_reduce<PCR_PLANE>, grid, block, 0, st, a); break;
    case PCR_VPLANE: hipLaunchKernelGGL(k_reduce<PCR_VPLANE>, grid, block, 0, st, a); break;
    default: hipLaunchKernelGGL(k_reduce<PCR_NDT>, grid, block, 0, st, a); break;
    }
}

void pcr_dev_launch_finalize(hipStream_t st, const FinArgs &f) {
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(1024), 0, st, f);
}

void pcr_dev_launch_coop(dim3 grid, hipStream_t st, const LinArgs &a) {
    hipLaunchKernelGGL((k_nn_coop<0>), grid, dim3(256), 0, st, a);
}

int pcr_dev_coop_blocks_per_cu() {
    int nb = 0;
    const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_nn_coop<0>, 256, 0);
    return (e == hipSuccess && nb > 0) ? nb : 4;
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
    if (t->gf.lbox && t->cs_h) hipLaunchKernelGGL((k_nn_counters<1, true>), dim3(a.nblocks), dim3(256), 0, ctx->stream, a, d.p);
    else if (t->gf.lbox) hipLaunchKernelGGL((k_nn_counters<0, true>), dim3(a.nblocks), dim3(256), 0, ctx->stream, a, d.p);
    else if (t->cs_h) hipLaunchKernelGGL(k_nn_counters<1>, dim3(a.nblocks), dim3(256), 0, ctx->stream, a, d.p);
    else hipLaunchKernelGGL(k_nn_counters<0>, dim3(a.nblocks), dim3(256), 0, ctx->stream, a, d.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(h, d.p, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < 11; ++i) out[i] = (double)h[i];
    return PCR_OK;
}

