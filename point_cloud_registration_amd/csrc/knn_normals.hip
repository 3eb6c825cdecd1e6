// Exact k nearest neighbours over the target's cell grid and k-NN PCA normals.
//
//   pcr_knn_query               KDTree(data).query(points, k) of the reference (kdtree.py:18-65)
//   pcr_target_estimate_normals estimate_norm_with_tree (estimate_normals.py:27-87): k-NN of every
//                               target point, covariance of the neighbours, eigenvector of the
//                               smallest eigenvalue
//
// One lane = one query; ties are ordered by (distance, original index), the oracle's rule (orc_knn_brute_f32).
//   k <= 16   knn_collect: class counts over the 27-cell block bound the k-th distance, the points within the bound are
//             queued unordered in LDS ([slot][lane]: a wave's accesses to one slot hit 64 banks) and ranked; queries it
//             cannot answer (sparse neighbourhoods, overfull classes) take the ring search with the register list KnnReg
//   k >  16   the ring expansion of nn_device.h with a sorted list in LDS (KnnList), the k-th best distance as the bound
#include "eigen3.h"
#include "nn_device.h"

#define KNN_BLOCK 64
#define KNN_MAX_K 64
#ifndef KNN_BATCH
#define KNN_BATCH 4          // candidates fetched together (knn_scan_range)
#endif
#define KNN_BATCH_ KNN_BATCH

// (round 4: the original indices are no longer kept in LDS -- they only matter when two distances are EQUAL, and are then
// read from the point records; 8 instead of 12 bytes per slot and lane lets 20 instead of 13 one-wave blocks share a CU's
// LDS at k = 15)
struct KnnList {
    float *d;        // [k][KNN_BLOCK] squared distances, ascending
    uint32_t *j;     // [k][KNN_BLOCK] cell-sorted indices
    const PtF *pts;  // the records the indices refer to (original index in .w)
    int k, cnt, lane;
    __device__ __forceinline__ float &D(int s) { return d[s * KNN_BLOCK + lane]; }
    __device__ __forceinline__ uint32_t &J(int s) { return j[s * KNN_BLOCK + lane]; }
    __device__ __forceinline__ uint32_t O(int s) { return pt_orig(pts[j[s * KNN_BLOCK + lane]]); }
    __device__ __forceinline__ void offer(float dist2, uint32_t jj, uint32_t oo, float &kth, uint32_t &kth_o);
    template <typename F>
    __device__ __forceinline__ void for_each(F &&f) { for (int s = 0; s < cnt; ++s) f(s, D(s), J(s)); }
    static constexpr bool kQueued = false, kCollect = false;
    __device__ __forceinline__ void flush(float &, uint32_t &) {}
    __device__ __forceinline__ void flush_if_full(float &, uint32_t &) {}
};

// Round 5: the list in REGISTERS for k <= 16 (the reference's default is 15, plane_icp.py:14).  The LDS
// list costs a chain of dependent LDS round trips per accepted candidate (read slot, compare, write slot, next) -- and a wave
// runs that chain whenever ANY of its lanes accepts, i.e. for nearly every candidate -- and caps the CU at 20 one-wave blocks.
// Here an insertion is straight-line: the list is sorted, so "new < slot s" is monotone in s and every slot takes either itself,
// the new element, or its predecessor -- 16 x (compare + 4 selects), no memory.  Ties (equal float32 distances) are ordered by
// the ORIGINAL index, read from the records only when two distances are equal.
//
// Round 5, second step: accepted candidates are QUEUED (KNN_QCAP (distance, index) pairs per lane in LDS) and inserted when
// some lane's queue is nearly full.  A wave pays for the 16-slot insertion (~240 VALU instructions) whenever ANY lane accepts
// -- a third of its candidate steps -- whereas a drain inserts for all lanes at once.  KNN_QCAP=0 builds the direct insertion
// again.  Since the collect path (knn_collect, below) this list only serves the queries that path hands back.
#define KNN_REG_K 16
#ifndef KNN_QCAP
#define KNN_QCAP 12
#endif
struct KnnReg {
    float d[KNN_REG_K];
    uint32_t j[KNN_REG_K];
    const PtF *pts;
    int k, cnt;
    static constexpr bool kQueued = KNN_QCAP > 0;
    float *qd;          // [KNN_QCAP][KNN_BLOCK] queued squared distances
    uint32_t *qj;       // [KNN_QCAP][KNN_BLOCK] their cell-sorted indices
    int qn;
    __device__ __forceinline__ void init(int k_, const PtF *pts_, char *smem) {
        k = k_; cnt = 0; pts = pts_; qn = 0;
        qd = (float *)smem + threadIdx.x;
        qj = (uint32_t *)(smem + sizeof(float) * (KNN_QCAP > 0 ? KNN_QCAP : 1) * KNN_BLOCK) + threadIdx.x;
#pragma unroll
        for (int s = 0; s < KNN_REG_K; ++s) { d[s] = __int_as_float(0x7f800000); j[s] = PCR_NONE; }
    }
    template <typename F>
    __device__ __forceinline__ void for_each(F &&f) const {            // nearest first
#pragma unroll
        for (int s = 0; s < KNN_REG_K; ++s) if (s < cnt) f(s, d[s], j[s]);
    }
    // exact insertion of one candidate (kth / kth_o: the k-th best once the list is full; before that kth may hold an
    // upper bound on it, which insertion leaves alone)
    __device__ __forceinline__ void insert(float dist2, uint32_t jj, float &kth, uint32_t &kth_o) {
        if (cnt == k) {
            bool acc = dist2 < kth;
            if (dist2 == kth) acc = pt_orig(pts[jj]) < pt_orig(pts[kth_o]);
            if (!acc) return;
        }
        float cd = dist2;                 // carried element: the new one until it is placed, then the slot's old content
        uint32_t cj = jj;
#pragma unroll
        for (int s = 0; s < KNN_REG_K; ++s) {
            const float ds = d[s];
            const uint32_t js = j[s];
            bool ins = dist2 < ds;
            if (dist2 == ds && js != PCR_NONE) ins = pt_orig(pts[jj]) < pt_orig(pts[js]);   // (an exact tie: the smaller original index first)
            d[s] = ins ? cd : ds; j[s] = ins ? cj : js;
            cd = ins ? ds : cd; cj = ins ? js : cj;
        }
        if (cnt < k) ++cnt;
        if (cnt == k) {
#pragma unroll
            for (int s = 0; s < KNN_REG_K; ++s) if (s == k - 1) { kth = d[s]; kth_o = j[s]; }
        }
    }
    __device__ __forceinline__ void flush(float &kth, uint32_t &kth_o) {
        if (KNN_QCAP > 0) {
            for (int s = 0; s < KNN_QCAP; ++s) {
                if (!__any(s < qn)) break;
                if (s < qn) insert(qd[s * KNN_BLOCK], qj[s * KNN_BLOCK], kth, kth_o);
            }
            qn = 0;
        }
    }
    // (called between batches of KNN_BATCH offers: a lane's queue never overflows)
    __device__ __forceinline__ void flush_if_full(float &kth, uint32_t &kth_o) {
        if (KNN_QCAP > 0 && __any(qn > KNN_QCAP - KNN_BATCH_)) flush(kth, kth_o);
    }
    __device__ __forceinline__ void offer(float dist2, uint32_t jj, uint32_t oo, float &kth, uint32_t &kth_o) {
        if (KNN_QCAP > 0) {
            // (cnt and kth are as of the last drain: accepting too much is harmless, insert() decides)
            bool acc = cnt < k ? !(dist2 > kth) : dist2 <= kth;
            if (acc) { qd[qn * KNN_BLOCK] = dist2; qj[qn * KNN_BLOCK] = jj; ++qn; }
        } else {
            (void)oo;
            insert(dist2, jj, kth, kth_o);
        }
    }
};

__device__ __forceinline__ void knn_offer(KnnList &L, float dist2, uint32_t jj, uint32_t oo, float &kth, uint32_t &kth_o) {
    // (kth_o holds the cell-sorted INDEX of the current k-th best; its original index is looked up only on an exact tie)
    if (L.cnt == L.k && !(dist2 < kth || (dist2 == kth && oo < pt_orig(L.pts[kth_o])))) return;
    int p = L.cnt < L.k ? L.cnt : L.k - 1;
    while (p > 0) {
        const float dp = L.D(p - 1);
        if (!(dp > dist2 || (dp == dist2 && L.O(p - 1) > oo))) break;      // (the original index only on a tie)
        L.D(p) = dp; L.J(p) = L.J(p - 1);
        --p;
    }
    L.D(p) = dist2; L.J(p) = jj;
    if (L.cnt < L.k) ++L.cnt;
    if (L.cnt == L.k) { kth = L.D(L.k - 1); kth_o = L.J(L.k - 1); }
}

__device__ __forceinline__ void KnnList::offer(float dist2, uint32_t jj, uint32_t oo, float &kth, uint32_t &kth_o) {
    knn_offer(*this, dist2, jj, oo, kth, kth_o);
}

// Candidates are fetched KNN_BATCH at a time (independent loads in flight before the first compare: the search is a chain
// of dependent round trips, one per candidate before round 4 -- k_knn_normals 1.49 ms per 1.06 M points at 5 % of the VALU
// rate).  A batch may read past the end of the range: those records exist (the array carries PCR_PTS_PAD sentinels behind
// its last point) but are NOT offered -- unlike in the 1-NN search a point of a later cell offered twice would sit in the
// list twice.
template <typename LIST>
__device__ __forceinline__ void knn_scan_range(LIST &L, const PtF *__restrict__ pts, uint32_t s, uint32_t e,
                                               float qx, float qy, float qz, float &kth, uint32_t &kth_o) {
    for (uint32_t j = s; j < e; j += KNN_BATCH) {
        PtF p[KNN_BATCH];
#pragma unroll
        for (int u = 0; u < KNN_BATCH; ++u) p[u] = pts[j + u];
#pragma unroll
        for (int u = 0; u < KNN_BATCH; ++u) {
            if (j + u < e) {
                const float dx = qx - p[u].x, dy = qy - p[u].y, dz = qz - p[u].z;
                const float d = dist2_f32(dx, dy, dz);
                L.offer(d, j + u, pt_orig(p[u]), kth, kth_o);
            }
        }
        L.flush_if_full(kth, kth_o);
    }
}

template <typename LIST>
__device__ __forceinline__ void knn_search(const Geom<float> &g, const PtF *__restrict__ pts,
                                           const uint32_t *__restrict__ cs, float qx, float qy, float qz, LIST &L,
                                           float kth_init = __int_as_float(0x7f800000)) {
    const float INF = __int_as_float(0x7f800000);
    float kth = kth_init;                 // (an upper bound on the k-th distance, when the caller has one)
    uint32_t kth_o = PCR_NONE;
    L.cnt = 0;
    const float lim = 1.0e9f;
    float rx = (qx - g.ox) * g.inv_h, ry = (qy - g.oy) * g.inv_h, rz = (qz - g.oz) * g.inv_h;
    rx = fminf(fmaxf(rx, -lim), lim); ry = fminf(fmaxf(ry, -lim), lim); rz = fminf(fmaxf(rz, -lim), lim);
    const int cx = (int)floorf(rx), cy = (int)floorf(ry), cz = (int)floorf(rz);
    const float fx = (qx - g.ox) - (float)cx * g.h, fy = (qy - g.oy) - (float)cy * g.h, fz = (qz - g.oz) - (float)cz * g.h;
    const float fmin_ = fminf(fminf(fminf(fx, g.h - fx), fminf(fy, g.h - fy)), fminf(fz, g.h - fz));
    const int k0 = max(max(max(-cx, cx - (g.nx - 1)), max(-cy, cy - (g.ny - 1))), max(max(-cz, cz - (g.nz - 1)), 0));
    const int kmax = max(max(max(cx, g.nx - 1 - cx), max(cy, g.ny - 1 - cy)), max(cz, g.nz - 1 - cz));
    for (int k = k0; k <= kmax; ++k) {
        if (k >= 2) L.flush(kth, kth_o);            // (the exact k-th distance before a ring beyond the block is opened)
        if (k >= 1) {
            const float lb = (float)(k - 1) * g.h + fmin_ - g.slack;
            if (lb > 0.f && lb * lb > kth) break;
        }
        const int zlo = max(cz - k, 0), zhi = min(cz + k, g.nz - 1);
        const int ylo = max(cy - k, 0), yhi = min(cy + k, g.ny - 1);
        for (int z = zlo; z <= zhi; ++z) {
            const int dzc = z - cz;
            float dzm = dzc == 0 ? 0.f : (dzc > 0 ? (float)dzc * g.h - fz : (float)(-dzc - 1) * g.h + fz);
            dzm = fmaxf(dzm - g.slack, 0.f);
            const float dz2 = dzm * dzm;
            if (dz2 > kth) continue;
            const bool zshell = (dzc == k) || (dzc == -k);
            for (int y = ylo; y <= yhi; ++y) {
                const int dyc = y - cy;
                float dym = dyc == 0 ? 0.f : (dyc > 0 ? (float)dyc * g.h - fy : (float)(-dyc - 1) * g.h + fy);
                dym = fmaxf(dym - g.slack, 0.f);
                const float dyz2 = dz2 + dym * dym;
                if (dyz2 > kth) continue;
                const size_t row = ((size_t)z * (size_t)g.ny + (size_t)y) * (size_t)g.nx;
                if (zshell || dyc == k || dyc == -k) {
                    int xl = max(cx - k, 0), xh = min(cx + k, g.nx - 1);
                    if (kth < INF) {
                        const float xr = __builtin_sqrtf(kth - dyz2) + g.slack;
                        const float a = (qx - xr - g.ox) * g.inv_h, b = (qx + xr - g.ox) * g.inv_h;
                        if (a > (float)xl) xl = (int)floorf(fminf(a, lim));
                        if (b < (float)xh) xh = (int)floorf(fmaxf(b, -lim));
                    }
                    if (xl <= xh) knn_scan_range(L, pts, cs[row + xl] & g.cs_mask, cs[row + xh + 1] & g.cs_mask, qx, qy, qz, kth, kth_o);
                } else {
                    const int xa = cx - k, xb = cx + k;
                    if (xa >= 0 && xa < g.nx) {
                        const float dxm = fmaxf((float)(k - 1) * g.h + fx - g.slack, 0.f);
                        if (dyz2 + dxm * dxm <= kth) knn_scan_range(L, pts, cs[row + xa] & g.cs_mask, cs[row + xa + 1] & g.cs_mask, qx, qy, qz, kth, kth_o);
                    }
                    if (xb >= 0 && xb < g.nx) {
                        const float dxm = fmaxf((float)k * g.h - fx - g.slack, 0.f);
                        if (dyz2 + dxm * dxm <= kth) knn_scan_range(L, pts, cs[row + xb] & g.cs_mask, cs[row + xb + 1] & g.cs_mask, qx, qy, qz, kth, kth_o);
                    }
                }
            }
        }
    }
    L.flush(kth, kth_o);
}

extern __shared__ __attribute__((aligned(16))) char knn_smem[];

// ---- the collect path (round 5; k <= 16) ---------------------------------------------------------------------------------
// Measured on the 1.06 M-point street cloud: without the sorted insertion the whole search takes 0.15 of k_knn_normals'
// 0.96 ms; 90 % of the points have their k = 15 neighbours within one cell edge, ~68 candidates each.  So, per query:
//   sweep 1   count the points of the 27-cell block around the query by squared distance: 16 classes, four per binade, the
//             largest (1.5 cell)^2, in per-lane LDS counters.  B = the upper edge of the first class at which the count
//             reaches k: the k-th neighbour is no farther.
//   sweep 2   if the ball of radius sqrt(B) lies inside the block, every point within B is IN the block: walk its rows again
//             (x ranges cut by B) and queue the points with d2 <= B -- k <= m <~ 1.2 k of them -- in LDS, no ordering.
//   rank      an entry's place = the number of queued entries before it in (distance, original index) order: m^2 trivial
//             compares, four entries per pass over the queue; places below k are the neighbours, nearest first.
// A query without a bound (fewer than k points within 1.5 cells: 9 % of a street scene), with a ball that leaves the block, or
// with more than KNN_C points inside B takes the ring search with the register list, started from B, and leaves its result
// in the same LDS arrays.  Results are identical to the list search's (test_fuzz_knn, g6, g7).
// Measured, 1.06 M points, k = 15: 0.92 -> 0.81 ms; the collect path alone takes 0.42 ms, the rest is those 9 %: each walks
// the rows of four rings, two dependent loads per row, and a wave lives as long as that chain.  Moving them into a second
// kernel -- compacted one per lane (0.75 ms on its own: too few waves to hide the chains), a wider 7^3 block sweep, a wave per
// query (0.60 ms: 2500 instructions per query, most lanes idle) -- lost every time: docs/EXPERIMENTS.md.
#ifndef KNN_C
#define KNN_C 24
#endif
#define KNN_NCLASS 16
static_assert(KNN_C >= KNN_REG_K && KNN_C >= KNN_NCLASS + 1 && KNN_C >= KNN_QCAP, "the LDS arrays are shared");
#define KNN_COLLECT_BYTES ((2 * sizeof(float) * KNN_C + KNN_REG_K) * KNN_BLOCK)

struct KnnOut {
    float *qd;          // [KNN_C][KNN_BLOCK] squared distances of the queued points (+inf: free slot)
    uint32_t *qj;       // [KNN_C][KNN_BLOCK] their cell-sorted indices; sweep 1 keeps its class counters here
    uint8_t *ord;       // [KNN_REG_K][KNN_BLOCK] queue slot of the r-th nearest
    int cnt;
    static constexpr bool kCollect = true;
    __device__ __forceinline__ void init(char *smem) {
        qd = (float *)smem + threadIdx.x;
        qj = (uint32_t *)(smem + sizeof(float) * KNN_C * KNN_BLOCK) + threadIdx.x;
        ord = (uint8_t *)(smem + 2 * sizeof(float) * KNN_C * KNN_BLOCK) + threadIdx.x;
        cnt = 0;
    }
    template <typename F>
    __device__ __forceinline__ void for_each(F &&f) const {            // nearest first
        for (int r = 0; r < cnt; ++r) {
            const int i = ord[r * KNN_BLOCK];
            f(r, qd[i * KNN_BLOCK], qj[i * KNN_BLOCK]);
        }
    }
};

// Rows of the 27-cell block around the query, cut to the ball of squared radius Bp (+inf: whole rows): f(s, e) = the range of
// records of one row.
template <typename F>
__device__ __forceinline__ void knn_rows(const Geom<float> &g, const uint32_t *__restrict__ cs, float qx, int cx, int cy, int cz,
                                         float fy, float fz, float Bp, F &&f) {
    const float INF = __int_as_float(0x7f800000), lim = 1.0e9f;
    const int xl = max(cx - 1, 0), xh = min(cx + 1, g.nx - 1);
    for (int z = max(cz - 1, 0); z <= min(cz + 1, g.nz - 1); ++z) {
        const int dzc = z - cz;
        float dzm = dzc == 0 ? 0.f : (dzc > 0 ? g.h - fz : fz);
        dzm = fmaxf(dzm - g.slack, 0.f);
        for (int y = max(cy - 1, 0); y <= min(cy + 1, g.ny - 1); ++y) {
            const int dyc = y - cy;
            float dym = dyc == 0 ? 0.f : (dyc > 0 ? g.h - fy : fy);
            dym = fmaxf(dym - g.slack, 0.f);
            const float dyz2 = dzm * dzm + dym * dym;
            int x0 = xl, x1 = xh;
            if (Bp < INF) {
                if (dyz2 > Bp) continue;
                const float xr = __builtin_sqrtf(Bp - dyz2) + g.slack;
                const float a = (qx - xr - g.ox) * g.inv_h, b = (qx + xr - g.ox) * g.inv_h;
                if (a > (float)x0) x0 = (int)floorf(fminf(a, lim));
                if (b < (float)x1) x1 = (int)floorf(fmaxf(b, -lim));
                if (x0 > x1) continue;
            }
            const size_t row = ((size_t)z * (size_t)g.ny + (size_t)y) * (size_t)g.nx;
            f(cs[row + x0] & g.cs_mask, cs[row + x1 + 1] & g.cs_mask);
        }
    }
}

__device__ __forceinline__ void knn_collect(const Geom<float> &g, const PtF *__restrict__ pts, const uint32_t *__restrict__ cs,
                                            float qx, float qy, float qz, int k, KnnOut &O) {
    const float INF = __int_as_float(0x7f800000);
    const float lim = 1.0e9f;
    float rx = (qx - g.ox) * g.inv_h, ry = (qy - g.oy) * g.inv_h, rz = (qz - g.oz) * g.inv_h;
    rx = fminf(fmaxf(rx, -lim), lim); ry = fminf(fmaxf(ry, -lim), lim); rz = fminf(fmaxf(rz, -lim), lim);
    const int cx = (int)floorf(rx), cy = (int)floorf(ry), cz = (int)floorf(rz);
    const bool inside = cx >= 0 && cx < g.nx && cy >= 0 && cy < g.ny && cz >= 0 && cz < g.nz;
    float B = INF;
    bool ok = false;
    int m = 0;
#pragma unroll
    for (int s = 0; s < KNN_C; ++s) O.qd[s * KNN_BLOCK] = INF;
    if (inside) {
        const float fx = (qx - g.ox) - (float)cx * g.h, fy = (qy - g.oy) - (float)cy * g.h, fz = (qz - g.oz) - (float)cz * g.h;
        // ---- sweep 1: class counts
        const int top = (int)(__float_as_uint(2.25f * g.h * g.h) >> 21), vlo = top - (KNN_NCLASS - 1);
#pragma unroll
        for (int b = 0; b <= KNN_NCLASS; ++b) O.qj[b * KNN_BLOCK] = 0u;
        knn_rows(g, cs, qx, cx, cy, cz, fy, fz, INF, [&](uint32_t s, uint32_t e) {
            for (uint32_t j = s; j < e; j += KNN_BATCH) {
                PtF p[KNN_BATCH];
#pragma unroll
                for (int u = 0; u < KNN_BATCH; ++u) p[u] = pts[j + u];
#pragma unroll
                for (int u = 0; u < KNN_BATCH; ++u) {
                    const float dx = qx - p[u].x, dy = qy - p[u].y, dz = qz - p[u].z;
                    const float d = dist2_f32(dx, dy, dz);
                    int b = (int)(__float_as_uint(d) >> 21) - vlo;               // (NaN / inf: beyond the last class)
                    b = j + u < e ? min(max(b, 0), KNN_NCLASS) : KNN_NCLASS;     // class KNN_NCLASS: not counted
                    atomicAdd(&O.qj[b * KNN_BLOCK], 1u);            // (ds_add_u32, the lane's own word: no read-modify-write chain)
                }
            }
        });
        unsigned cum = 0;
        for (int b = 0; b < KNN_NCLASS; ++b) {
            cum += O.qj[b * KNN_BLOCK];
            if (cum >= (unsigned)k) {                       // every counted d2 has bits >> 21 <= vlo + b: d2 < the float below
                if (vlo + b + 1 > 0 && vlo + b + 1 < 0x3fc) B = __uint_as_float((unsigned)(vlo + b + 1) << 21);
                break;
            }
        }
        // ---- sweep 2: queue the points within B, when they all lie in the block
        const float fmin_ = fminf(fminf(fminf(fx, g.h - fx), fminf(fy, g.h - fy)), fminf(fz, g.h - fz));
        const float lb = g.h + fmin_ - g.slack;             // nothing outside the block is closer (the ring search's bound)
        if (B < INF && lb > 0.f && B < lb * lb) {
            knn_rows(g, cs, qx, cx, cy, cz, fy, fz, B, [&](uint32_t s, uint32_t e) {
                for (uint32_t j = s; j < e; j += KNN_BATCH) {
                    PtF p[KNN_BATCH];
#pragma unroll
                    for (int u = 0; u < KNN_BATCH; ++u) p[u] = pts[j + u];
#pragma unroll
                    for (int u = 0; u < KNN_BATCH; ++u) {
                        const float dx = qx - p[u].x, dy = qy - p[u].y, dz = qz - p[u].z;
                        const float d = dist2_f32(dx, dy, dz);
                        if (j + u < e && d <= B) {
                            if (m < KNN_C) { O.qd[m * KNN_BLOCK] = d; O.qj[m * KNN_BLOCK] = j + u; }
                            ++m;
                        }
                    }
                }
            });
            ok = m <= KNN_C;
        }
    }
    if (!ok) {
        // ---- handed back: the ring search from the bound, its list copied out
        KnnReg L;
        L.init(k, pts, (char *)(O.qd - threadIdx.x));
        knn_search(g, pts, cs, qx, qy, qz, L, B);
#pragma unroll
        for (int s = 0; s < KNN_REG_K; ++s) {
            if (s < L.cnt) { O.qd[s * KNN_BLOCK] = L.d[s]; O.qj[s * KNN_BLOCK] = L.j[s]; O.ord[s * KNN_BLOCK] = (uint8_t)s; }
        }
        O.cnt = L.cnt;
    }
    // ---- rank (lanes of the collect path; the wave's trip counts are those of its fullest queue)
    const int mm = ok ? m : 0;
    for (int i0 = 0; i0 < KNN_C; i0 += 4) {
        if (!__any(i0 < mm)) break;
        float di[4];
        int lt[4] = {0, 0, 0, 0}, eq[4] = {0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < 4; ++u) di[u] = i0 + u < KNN_C ? O.qd[(i0 + u) * KNN_BLOCK] : INF;
        for (int s = 0; s < KNN_C; ++s) {
            if (!__any(s < mm)) break;
            const float ds = O.qd[s * KNN_BLOCK];                       // (+inf beyond the lane's own m: before nothing)
#pragma unroll
            for (int u = 0; u < 4; ++u) { lt[u] += ds < di[u]; eq[u] += ds == di[u]; }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u;
            if (i < mm) {
                int r = lt[u];
                if (eq[u] > 1) {                                        // equal distances: the smaller original index first
                    const uint32_t oi = pt_orig(pts[O.qj[i * KNN_BLOCK]]);
                    for (int s = 0; s < mm; ++s)
                        if (s != i && O.qd[s * KNN_BLOCK] == di[u] && pt_orig(pts[O.qj[s * KNN_BLOCK]]) < oi) ++r;
                }
                if (r < k) O.ord[r * KNN_BLOCK] = (uint8_t)i;
            }
        }
    }
    if (ok) O.cnt = min(m, k);
}

__device__ __forceinline__ KnnList knn_list(int k) {
    KnnList L;
    L.k = k; L.cnt = 0; L.lane = threadIdx.x;
    L.d = (float *)knn_smem;
    L.j = (uint32_t *)(knn_smem + sizeof(float) * k * KNN_BLOCK);
    L.pts = nullptr;
    return L;
}

template <typename LIST>
__device__ __forceinline__ void knn_query_finish(const PtF *pts, int64_t n_target, int64_t i, int k, float *dist, int64_t *idx, LIST &L) {
    for (int s = L.cnt; s < k; ++s) { dist[i * k + s] = __int_as_float(0x7f800000); idx[i * k + s] = n_target; }
    L.for_each([&](int s, float ds, uint32_t js) {
        dist[i * k + s] = __builtin_sqrtf(ds);
        idx[i * k + s] = (int64_t)pt_orig(pts[js]);
    });
}

// REG = 1: the collect path (k <= 16); 0: the LDS list
template <int REG>
__global__ void __launch_bounds__(KNN_BLOCK) k_knn_query(Geom<float> g, const PtF *pts, const uint32_t *cs, int64_t n_target,
                                                         const float *q, int64_t m, int k, float *dist, int64_t *idx) {
    const int64_t i = (int64_t)blockIdx.x * KNN_BLOCK + threadIdx.x;
    if (i >= m) return;
    if (REG) {
        KnnOut L;
        L.init(knn_smem);
        knn_collect(g, pts, cs, q[3 * i], q[3 * i + 1], q[3 * i + 2], k, L);
        knn_query_finish(pts, n_target, i, k, dist, idx, L);
    } else {
        KnnList L = knn_list(k);
        L.pts = pts;
        knn_search(g, pts, cs, q[3 * i], q[3 * i + 1], q[3 * i + 2], L);
        knn_query_finish(pts, n_target, i, k, dist, idx, L);
    }
}

template <typename LIST>
__device__ __forceinline__ void knn_normals_finish(const PtF *pts, const PtF me, int64_t i, int k, int compat, PtN *pn, LIST &L) {
    double c[6];
    if (compat) {
        // estimate_normals.py:56-72: float32 running sums over the neighbours (nearest first),
        // cov = E[pp^T] - mu mu^T in float32
        float sx = 0, sy = 0, sz = 0, xx = 0, xy = 0, xz = 0, yy = 0, yz = 0, zz = 0;
        L.for_each([&](int, float, uint32_t js) {
            const PtF p = pts[js];
            sx += p.x; sy += p.y; sz += p.z;
            xx += p.x * p.x; xy += p.x * p.y; xz += p.x * p.z; yy += p.y * p.y; yz += p.y * p.z; zz += p.z * p.z;
        });
        const float kf = (float)k;
        const float mx = sx / kf, my = sy / kf, mz = sz / kf;
        c[0] = xx / kf - mx * mx; c[1] = xy / kf - mx * my; c[2] = xz / kf - mx * mz;
        c[3] = yy / kf - my * my; c[4] = yz / kf - my * mz; c[5] = zz / kf - mz * mz;
    } else {
        double mx = 0, my = 0, mz = 0;
        L.for_each([&](int, float, uint32_t js) { const PtF p = pts[js]; mx += p.x; my += p.y; mz += p.z; });
        const double kd = (double)(L.cnt > 0 ? L.cnt : 1);
        mx /= kd; my /= kd; mz /= kd;
#pragma unroll
        for (int a = 0; a < 6; ++a) c[a] = 0;
        L.for_each([&](int, float, uint32_t js) {
            const PtF p = pts[js];
            const double dx = p.x - mx, dy = p.y - my, dz = p.z - mz;
            c[0] += dx * dx; c[1] += dx * dy; c[2] += dx * dz; c[3] += dy * dy; c[4] += dy * dz; c[5] += dz * dz;
        });
#pragma unroll
        for (int a = 0; a < 6; ++a) c[a] /= kd;
    }
    double nv[3];
    smallest_eigvec3(c, nv);
    PtN r;                                      // the PlaneICP gather record: point + normal in 32 bytes
    r.x = me.x; r.y = me.y; r.z = me.z; r.orig = pt_orig(me);
    r.nx = (float)nv[0]; r.ny = (float)nv[1]; r.nz = (float)nv[2]; r.pad = 0;
    pn[i] = r;
}

// points in the 27-cell block around a position (from cell_start alone)
#ifndef KNN_SPARSE_T
#define KNN_SPARSE_T 30
#endif
__device__ __forceinline__ uint32_t knn_block_count(const Geom<float> &g, const uint32_t *__restrict__ cs, float qx, float qy, float qz) {
    const float lim = 1.0e9f;
    const int cx = (int)floorf(fminf(fmaxf((qx - g.ox) * g.inv_h, -lim), lim)), cy = (int)floorf(fminf(fmaxf((qy - g.oy) * g.inv_h, -lim), lim)),
              cz = (int)floorf(fminf(fmaxf((qz - g.oz) * g.inv_h, -lim), lim));
    if (!(cx >= 0 && cx < g.nx && cy >= 0 && cy < g.ny && cz >= 0 && cz < g.nz)) return 0;
    const int xl = max(cx - 1, 0), xh = min(cx + 1, g.nx - 1);
    uint32_t t = 0;
    for (int z = max(cz - 1, 0); z <= min(cz + 1, g.nz - 1); ++z)
        for (int y = max(cy - 1, 0); y <= min(cy + 1, g.ny - 1); ++y) {
            const size_t row = ((size_t)z * (size_t)g.ny + (size_t)y) * (size_t)g.nx;
            t += (cs[row + xh + 1] & g.cs_mask) - (cs[row + xl] & g.cs_mask);
        }
    return t;
}

// Sparse neighbourhoods first: a wave of such points lives several times as long as the others (ring search over hundreds of
// rows), and the cell-sorted order tends to keep them together at one end of the array -- at the far end they were the kernel's
// tail (1.06 M-point street cloud: 0.82 ms in array order, 0.71 reversed).  order[] = the blocks whose FIRST point has fewer
// than KNN_SPARSE_T points in its 27-cell block, then the others (cnt: two zeroed counters).
__global__ void __launch_bounds__(256) k_knn_order(Geom<float> g, const PtF *pts, const uint32_t *cs, uint32_t nb, uint32_t *order,
                                                   uint32_t *cnt) {
    const uint32_t b = blockIdx.x * 256u + threadIdx.x;
    const bool valid = b < nb;
    bool sparse = false;
    if (valid) {
        const PtF p0 = pts[(int64_t)b * KNN_BLOCK];
        sparse = knn_block_count(g, cs, p0.x, p0.y, p0.z) < KNN_SPARSE_T;
    }
    const int lane = threadIdx.x & 63;
    const unsigned long long ms = __ballot(valid && sparse), md = __ballot(valid && !sparse);
    uint32_t bs = 0, bd = 0;
    if (lane == 0) { if (ms) bs = atomicAdd(&cnt[0], (uint32_t)__popcll(ms)); if (md) bd = atomicAdd(&cnt[1], (uint32_t)__popcll(md)); }
    bs = __shfl(bs, 0, 64); bd = __shfl(bd, 0, 64);
    const unsigned long long below = (1ull << lane) - 1ull;
    if (valid && sparse) order[bs + (uint32_t)__popcll(ms & below)] = b;
    if (valid && !sparse) order[nb - 1u - (bd + (uint32_t)__popcll(md & below))] = b;
}

// normals of the target's own points, processed (and written) in cell-sorted order
template <int REG>
__global__ void __launch_bounds__(KNN_BLOCK) k_knn_normals(Geom<float> g, const PtF *pts, const uint32_t *cs, int64_t n,
                                                           int k, int compat, PtN *pn, const uint32_t *__restrict__ order) {
    const int64_t i = (int64_t)(order ? order[blockIdx.x] : blockIdx.x) * KNN_BLOCK + threadIdx.x;
    if (i >= n) return;
    const PtF me = pts[i];
    if (REG) {
        KnnOut L;
        L.init(knn_smem);
        knn_collect(g, pts, cs, me.x, me.y, me.z, k, L);
        knn_normals_finish(pts, me, i, k, compat, pn, L);
    } else {
        KnnList L = knn_list(k);
        L.pts = pts;
        knn_search(g, pts, cs, me.x, me.y, me.z, L);
        knn_normals_finish(pts, me, i, k, compat, pn, L);
    }
}

// k <= 16: the register list (PCR_KNN_REG=0: the LDS list for every k, for A/B)
static bool knn_use_registers(int k) {
    static const int allow = getenv("PCR_KNN_REG") ? atoi(getenv("PCR_KNN_REG")) : 1;
    return allow != 0 && k <= KNN_REG_K;
}

static pcr_status check_k(int k) {
    if (k < 1 || k > KNN_MAX_K) { pcr_set_error("k must be in [1, %d]", KNN_MAX_K); return PCR_ERR_INVALID; }
    return PCR_OK;
}

extern "C" pcr_status pcr_knn_query(pcr_target *t, const float *q, int64_t m, int k, float *dist, int64_t *idx) {
    PCR_REQUIRE(t && (q || m == 0) && (dist || m == 0) && (idx || m == 0), "NULL argument");
    PCR_REQUIRE(!t->is_voxel, "k-NN needs a point target");
    PCR_TRY(check_k(k));
    pcr_context *ctx = t->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    if (m == 0) return PCR_OK;
    CtxScope scope(ctx);
    DevBuf<float> d_q, d_dist;
    DevBuf<int64_t> d_idx;
    HIP_TRY(d_q.alloc(3 * (size_t)m));
    HIP_TRY(d_dist.alloc((size_t)m * k));
    HIP_TRY(d_idx.alloc((size_t)m * k));
    HIP_TRY(hipMemcpyAsync(d_q.p, q, 12 * (size_t)m, hipMemcpyHostToDevice, ctx->stream));
    const Geom<float> &g = t->gf;
    const size_t smem = 2 * sizeof(float) * (size_t)k * KNN_BLOCK;
    const dim3 qgrid((unsigned)((m + KNN_BLOCK - 1) / KNN_BLOCK));
    if (knn_use_registers(k))
        hipLaunchKernelGGL(k_knn_query<1>, qgrid, dim3(KNN_BLOCK), KNN_COLLECT_BYTES, ctx->stream,
                           g, t->pts, t->cell_start, t->n, (const float *)d_q.p, m, k, d_dist.p, d_idx.p);
    else
        hipLaunchKernelGGL(k_knn_query<0>, qgrid, dim3(KNN_BLOCK), smem, ctx->stream,
                           g, t->pts, t->cell_start, t->n, (const float *)d_q.p, m, k, d_dist.p, d_idx.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(dist, d_dist.p, 4 * (size_t)m * k, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(idx, d_idx.p, 8 * (size_t)m * k, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return PCR_OK;
}

extern "C" pcr_status pcr_target_estimate_normals(pcr_target *t, int k, int compat, float *normals_out) {
    PCR_REQUIRE(t, "NULL argument");
    PCR_REQUIRE(!t->is_voxel, "normals belong to point targets");
    PCR_TRY(check_k(k));
    pcr_context *ctx = t->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    CtxScope scope(ctx);
    if (!t->pn) HIP_TRY(pcr_persist_alloc((void **)&t->pn, sizeof(PtN) * (size_t)(t->n ? t->n : 1)));
    if (t->n > 0) {
        const Geom<float> &g = t->gf;
        const size_t smem = 2 * sizeof(float) * (size_t)k * KNN_BLOCK;
        const dim3 ngrid((unsigned)((t->n + KNN_BLOCK - 1) / KNN_BLOCK));
        static const int sparse_first = getenv("PCR_KNN_SPARSE_FIRST") ? atoi(getenv("PCR_KNN_SPARSE_FIRST")) : 1;
        if (knn_use_registers(k)) {
            DevBuf<uint32_t> order, cnt;
            // (only where the tail is a visible share of the kernel: at 1e7 / 1e8 points the reordering costs 3 % in locality)
            if (sparse_first && ngrid.x > 1024 && ngrid.x <= 65536) {
                HIP_TRY(order.alloc(ngrid.x)); HIP_TRY(cnt.alloc(2));
                HIP_TRY(hipMemsetAsync(cnt.p, 0, 2 * sizeof(uint32_t), ctx->stream));
                hipLaunchKernelGGL(k_knn_order, dim3((ngrid.x + 255) / 256), dim3(256), 0, ctx->stream, g, t->pts, t->cell_start, ngrid.x, order.p, cnt.p);
            }
            hipLaunchKernelGGL(k_knn_normals<1>, ngrid, dim3(KNN_BLOCK), KNN_COLLECT_BYTES, ctx->stream, g, t->pts, t->cell_start, t->n, k, compat, t->pn,
                               (const uint32_t *)order.p);
        } else {
            hipLaunchKernelGGL(k_knn_normals<0>, ngrid, dim3(KNN_BLOCK), smem, ctx->stream, g, t->pts, t->cell_start, t->n, k, compat, t->pn, (const uint32_t *)nullptr);
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    if (normals_out) return pcr_target_get_normals(t, normals_out);
    return PCR_OK;
}
