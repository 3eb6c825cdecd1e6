// GPU build of a voxel target = VoxelGrid.set_points + calc_icov of the reference
// (voxel.py:104-165, 69-102), behind pcr_target_voxels_create.
//
//   keys      k_voxel_keys     integer hash of floor(p / voxel_size), evaluated in the dtype of the
//                              cloud (voxel.py:12-21; bit-exact: integer work)
//   grouping  rocPRIM radix sort of (key, point index) + run-length encode: voxels come out in
//             ascending key order (np.unique) and, the sort being stable, every voxel's points in
//             ascending point index (np.bincount's accumulation order)
//   stats     k_voxel_stats    one wave per voxel: float64 mean, two-pass sample covariance / max(n-1, 1)
//             k_voxel_eig      one lane per kept voxel: smallest-eigenvector normal, closed-form inverse
//   index     pcr_voxel_target_finish: dense grid over the kept centroids (float64 search)
#include <hipcub/hipcub.hpp>

#include "eigen3.h"
#include "pcr_internal.h"

pcr_status pcr_voxel_target_finish(pcr_context *ctx, pcr_target *t, double voxel_size);   // api.hip

__device__ __forceinline__ long long pymod(long long a, long long m) {
    const long long r = a % m;
    return r < 0 ? r + m : r;
}

// (bias: added to every key so that the radix sort only has to look at the bits keys actually use -- see voxel_build)
template <typename T>
__global__ void __launch_bounds__(256) k_voxel_keys(const T *__restrict__ xyz, int64_t n, T voxel_size, long long bias,
                                                    long long *keys, uint32_t *idx) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long long P = 116101LL, M = 10000000000LL;
    // floor(points / voxel_size).astype(int64) in the array's own precision (voxel.py:16)
    const long long x = (long long)floor(xyz[3 * i] / voxel_size);
    const long long y = (long long)floor(xyz[3 * i + 1] / voxel_size);
    const long long z = (long long)floor(xyz[3 * i + 2] / voxel_size);
    keys[i] = pymod((pymod(z * P, M) + y) * P, M) + x + bias;                     // voxel.py:20
    idx[i] = (uint32_t)i;
}

__global__ void __launch_bounds__(256) k_keep_flags(const uint32_t *__restrict__ counts, int64_t nu, int min_points,
                                                    uint32_t *flags) {
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (v < nu) flags[v] = counts[v] >= (uint32_t)min_points ? 1u : 0u;
    if (v == nu) flags[v] = 0u;
}

// One WAVE per voxel.  The voxel's points are gathered cooperatively (64 at a time) into LDS as float64;
// the sums themselves stay strictly sequential in point order -- np.bincount's accumulation order, which
// makes mean / covariance bit-identical to the oracle -- but every component is an independent sum, so
// lanes 0..2 run the three mean sums and lanes 0..5 the six covariance sums side by side.  (One LANE per
// voxel, the first version, walked ~50 scattered points twice per lane: 11 ms for 1.06 M points.)
#define VS_CHUNK 64
template <typename T>
__global__ void __launch_bounds__(256) k_voxel_stats(const T *__restrict__ xyz, const uint32_t *__restrict__ order,
                                                     const long long *__restrict__ ukeys,
                                                     const uint32_t *__restrict__ counts,
                                                     const uint32_t *__restrict__ seg_start,
                                                     const uint32_t *__restrict__ keep_pos, int64_t nu, int min_points,
                                                     long long key_bias, double *mean, double *cov,
                                                     int64_t *out_counts, int64_t *out_keys) {
    __shared__ double buf[4][3][VS_CHUNK];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t v = (int64_t)blockIdx.x * 4 + wave;
    if (v >= nu) return;
    const uint32_t cnt = counts[v];
    if (cnt < (uint32_t)min_points) return;                                        // voxel.py:151
    const uint32_t s = seg_start[v], o = keep_pos[v];
    double (*b)[VS_CHUNK] = buf[wave];
    // pass 1: mean (lanes 0..2: x, y, z), points in ascending index order
    double acc = 0.0;
    for (uint32_t t0 = 0; t0 < cnt; t0 += VS_CHUNK) {
        const uint32_t m = cnt - t0 < VS_CHUNK ? cnt - t0 : VS_CHUNK;
        if ((uint32_t)lane < m) {
            const size_t i = order[s + t0 + lane];
            b[0][lane] = (double)xyz[3 * i]; b[1][lane] = (double)xyz[3 * i + 1]; b[2][lane] = (double)xyz[3 * i + 2];
        }
        __builtin_amdgcn_wave_barrier();
        if (lane < 3) for (uint32_t t = 0; t < m; ++t) acc += b[lane][t];
        __builtin_amdgcn_wave_barrier();
    }
    const double mcomp = acc / (double)cnt;                                        // voxel.py:118-121 (lanes 0..2)
    const double mx = __shfl(mcomp, 0, 64), my = __shfl(mcomp, 1, 64), mz = __shfl(mcomp, 2, 64);
    // pass 2: covariance (lanes 0..5: xx xy xz yy yz zz)
    const int ia = lane == 0 || lane == 1 || lane == 2 ? 0 : (lane == 3 || lane == 4 ? 1 : 2);
    const int ib = lane == 0 ? 0 : (lane == 1 || lane == 3 ? 1 : 2);
    const double ma = ia == 0 ? mx : (ia == 1 ? my : mz), mb = ib == 0 ? mx : (ib == 1 ? my : mz);
    double c = 0.0;
    for (uint32_t t0 = 0; t0 < cnt; t0 += VS_CHUNK) {
        const uint32_t m = cnt - t0 < VS_CHUNK ? cnt - t0 : VS_CHUNK;
        if ((uint32_t)lane < m) {
            const size_t i = order[s + t0 + lane];
            b[0][lane] = (double)xyz[3 * i]; b[1][lane] = (double)xyz[3 * i + 1]; b[2][lane] = (double)xyz[3 * i + 2];
        }
        __builtin_amdgcn_wave_barrier();
        if (lane < 6) for (uint32_t t = 0; t < m; ++t) c += (b[ia][t] - ma) * (b[ib][t] - mb);
        __builtin_amdgcn_wave_barrier();
    }
    const double den = (double)(cnt > 2 ? cnt - 1 : 1);                            // max(n-1, 1), voxel.py:136
    c /= den;
    double c6[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) c6[a] = __shfl(c, a, 64);
    if (lane != 0) return;
    mean[3 * (size_t)o] = mx; mean[3 * (size_t)o + 1] = my; mean[3 * (size_t)o + 2] = mz;
    double m9[9] = {c6[0], c6[1], c6[2], c6[1], c6[3], c6[4], c6[2], c6[4], c6[5]};
#pragma unroll
    for (int a = 0; a < 9; ++a) cov[9 * (size_t)o + a] = m9[a];
    out_counts[o] = (int64_t)cnt;
    out_keys[o] = ukeys[v] - key_bias;
}

// normal (smallest eigenvector, voxel.py:157-158) and closed-form inverse (voxel.py:69-102) of every kept voxel's covariance:
// one LANE per voxel.  (Round 5: k_voxel_stats used to end with these -- a few hundred float64 instructions on lane 0 of a
// wave per voxel, most of the kernel's 116 us per 1.06 M points.)
__global__ void __launch_bounds__(256) k_voxel_eig(const double *__restrict__ cov, int64_t nk, double *norm, double *icov) {
    const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= nk) return;
    double m9[9];
#pragma unroll
    for (int a = 0; a < 9; ++a) m9[a] = cov[9 * (size_t)o + a];
    const double c6[6] = {m9[0], m9[1], m9[2], m9[4], m9[5], m9[8]};
    double nv[3];
    smallest_eigvec3(c6, nv);
    norm[3 * (size_t)o] = nv[0]; norm[3 * (size_t)o + 1] = nv[1]; norm[3 * (size_t)o + 2] = nv[2];
    double ic[9];
    icov_closed_form(m9, ic);
#pragma unroll
    for (int a = 0; a < 9; ++a) icov[9 * (size_t)o + a] = ic[a];
}

template <typename T>
static pcr_status voxel_build(pcr_context *ctx, const T *d_xyz, int64_t n, double voxel_size, int min_points,
                              pcr_target *t, float x_lo, float x_hi) {
    const size_t nn = (size_t)(n > 0 ? n : 1);
    const unsigned nb = (unsigned)((n + 255) / 256);
    // temporaries: blocks of the context's cache (DevBuf), gone on every exit path
    DevBuf<long long> k1, k2, ukeys;
    DevBuf<uint32_t> i1, i2, counts, seg, flags;
    DevBuf<int> d_runs;
    DevBuf<char> tmp;
    int64_t nu = 0, nk = 0;
    HIP_TRY(k1.alloc(nn)); HIP_TRY(k2.alloc(nn)); HIP_TRY(ukeys.alloc(nn));
    HIP_TRY(i1.alloc(nn)); HIP_TRY(i2.alloc(nn));
    HIP_TRY(counts.alloc(nn + 1)); HIP_TRY(seg.alloc(nn + 1)); HIP_TRY(flags.alloc(nn + 1));
    HIP_TRY(d_runs.alloc(1));
    // A key is (a residue in [0, 1e10)) + x with x = floor(px / voxel_size): adding bias = -min(x, 0) makes every key
    // non-negative and smaller than 1e10 + (x_max - x_min), i.e. ~34 significant bits for any real cloud, and a stable radix
    // sort of those bits orders them exactly as a sort of all 64 does: 5 onesweep passes instead of 8 (25 us each per 1.06 M
    // points).  The x range comes from the bounding box (float32-rounded: a margin of 2 voxels + 1e-6 relative covers it);
    // coordinates beyond what 60 bits hold keep the full sort.
    long long key_bias = 0;
    int key_bits = 64;
    {
        const double xl = floor((double)x_lo / voxel_size), xh = floor((double)x_hi / voxel_size);
        const double xmin = xl - 2.0 - 1e-6 * fabs(xl), xmax = xh + 2.0 + 1e-6 * fabs(xh);
        if (std::isfinite(xmin) && std::isfinite(xmax) && fabs(xmin) < 1e17 && fabs(xmax) < 1e17) {
            key_bias = xmin < 0 ? (long long)(-xmin) : 0;
            const double top = 1.0e10 + xmax + (double)key_bias;
            int b = 1;
            while (b < 62 && ldexp(1.0, b) <= top) ++b;
            key_bits = b < 62 ? b : 64;
            if (key_bits == 64) key_bias = 0;
        }
    }
    if (n > 0) {
        hipLaunchKernelGGL(k_voxel_keys<T>, dim3(nb), dim3(256), 0, ctx->stream, d_xyz, n, (T)voxel_size, key_bias, k1.p, i1.p);
        size_t tb = 0;
        HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, k1.p, k2.p, i1.p, i2.p, (int)n, 0, key_bits, ctx->stream));
        HIP_TRY(tmp.alloc_bytes(tb));
        HIP_TRY(hipcub::DeviceRadixSort::SortPairs(tmp.p, tb, k1.p, k2.p, i1.p, i2.p, (int)n, 0, key_bits, ctx->stream));
        tb = 0;
        HIP_TRY(hipcub::DeviceRunLengthEncode::Encode(nullptr, tb, k2.p, ukeys.p, counts.p, d_runs.p, (int)n, ctx->stream));
        DevBuf<char> tmp2;
        HIP_TRY(tmp2.alloc_bytes(tb));
        HIP_TRY(hipcub::DeviceRunLengthEncode::Encode(tmp2.p, tb, k2.p, ukeys.p, counts.p, d_runs.p, (int)n, ctx->stream));
        int runs = 0;
        HIP_TRY(hipMemcpyAsync(&runs, d_runs.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        nu = runs;
        // segment starts (exclusive scan of counts) and compacted positions of the kept voxels
        HIP_TRY(hipMemsetAsync(counts.p + nu, 0, 4, ctx->stream));
        tb = 0;
        HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, counts.p, seg.p, (int)nu + 1, ctx->stream));
        DevBuf<char> tmp3;
        HIP_TRY(tmp3.alloc_bytes(tb));
        HIP_TRY(hipcub::DeviceScan::ExclusiveSum(tmp3.p, tb, counts.p, seg.p, (int)nu + 1, ctx->stream));
        hipLaunchKernelGGL(k_keep_flags, dim3((unsigned)((nu + 256) / 256)), dim3(256), 0, ctx->stream, counts.p, nu, min_points, flags.p);
        HIP_TRY(hipcub::DeviceScan::ExclusiveSum(tmp3.p, tb, flags.p, flags.p, (int)nu + 1, ctx->stream));
        uint32_t kept = 0;
        HIP_TRY(hipMemcpyAsync(&kept, flags.p + nu, 4, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        nk = kept;
    }
    {
        const size_t kk = (size_t)(nk > 0 ? nk : 1);
        HIP_TRY(pcr_persist_alloc((void **)&t->st_mean, 8 * 3 * kk)); HIP_TRY(pcr_persist_alloc((void **)&t->st_cov, 8 * 9 * kk));
        HIP_TRY(pcr_persist_alloc((void **)&t->st_norm, 8 * 3 * kk)); HIP_TRY(pcr_persist_alloc((void **)&t->st_icov, 8 * 9 * kk));
        HIP_TRY(pcr_persist_alloc((void **)&t->st_counts, 8 * kk)); HIP_TRY(pcr_persist_alloc((void **)&t->st_keys, 8 * kk));
        if (nu > 0) {
            hipLaunchKernelGGL(k_voxel_stats<T>, dim3((unsigned)((nu + 3) / 4)), dim3(256), 0, ctx->stream, d_xyz, i2.p,
                               ukeys.p, counts.p, seg.p, flags.p, nu, min_points, key_bias, t->st_mean, t->st_cov,
                               t->st_counts, t->st_keys);
            if (nk > 0)
                hipLaunchKernelGGL(k_voxel_eig, dim3((unsigned)((nk + 255) / 256)), dim3(256), 0, ctx->stream, (const double *)t->st_cov, nk,
                                   t->st_norm, t->st_icov);
            HIP_TRY(hipGetLastError());          // (no synchronisation: the centroid grid is built on the same stream)
        }
        t->n = nk;
    }
    return pcr_voxel_target_finish(ctx, t, voxel_size);
}

extern "C" pcr_status pcr_target_voxels_create(pcr_context *ctx, const void *xyz, int xyz_is_f64, int64_t n,
                                               double voxel_size, int min_points, pcr_target **out) {
    PCR_REQUIRE(ctx && out, "NULL argument");
    PCR_REQUIRE(n >= 0 && (xyz || n == 0), "bad point array");
    PCR_REQUIRE(n < ((int64_t)1 << 31), "at most 2^31-1 points per target");
    PCR_REQUIRE(voxel_size > 0, "voxel_size must be positive");
    HIP_TRY(hipSetDevice(ctx->device));
    CtxScope scope(ctx);
    const size_t elem = xyz_is_f64 ? 8 : 4;
    DevBuf<char> d_xyz;
    HIP_TRY(d_xyz.alloc_bytes(elem * 3 * (size_t)(n > 0 ? n : 1)));
    if (n > 0) HIP_TRY(hipMemcpyAsync(d_xyz.p, xyz, elem * 3 * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    int64_t nonfinite = 0;
    float lo[3], hi[3];
    // (no wait behind the copy: the bounding box's read-back synchronises the stream, on every path out of it)
    {
        const pcr_status sb = pcr_count_nonfinite(ctx, d_xyz.p, xyz_is_f64, n, &nonfinite, lo, hi);
        if (sb != PCR_OK) { (void)hipStreamSynchronize(ctx->stream); return sb; }
    }
    if (nonfinite > 0) {               // floor(NaN / voxel_size).astype(int64) is undefined in the reference as well
        pcr_set_error("cloud has %lld point(s) with a non-finite coordinate; drop them first", (long long)nonfinite);
        return PCR_ERR_INVALID;
    }
    pcr_target *t = new pcr_target();
    t->ctx = ctx; t->is_voxel = 1; t->serial = ctx->next_serial++;
    pcr_status s = xyz_is_f64 ? voxel_build<double>(ctx, (const double *)d_xyz.p, n, voxel_size, min_points, t, lo[0], hi[0])
                              : voxel_build<float>(ctx, (const float *)d_xyz.p, n, voxel_size, min_points, t, lo[0], hi[0]);
    d_xyz.reset();
    if (s != PCR_OK) { pcr_target_destroy(t); return s; }
    *out = t;
    return PCR_OK;
}
