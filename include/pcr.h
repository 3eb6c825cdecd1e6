/*
 * pcr.h -- C ABI of the MI355X-native point-cloud registration core (libpcr_hip.so).
 *
 * The reference (scomup/point-cloud-registration) is pure Python/NumPy and has no FFI of
 * its own; its seams are Python duck types.  This header is the
 * drop-in boundary a maintainer would bind with ctypes from inside the reference's classes
 * (INTEGRATION.md shows the stub).  Every entry point names the reference interface it
 * replaces (paths relative to /root/reference/point_cloud_registration/).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, opaque handles; no C++/torch types.
 *   - every function returns a pcr_status (0 = PCR_OK); pcr_last_error() gives the message
 *     of the last failure on the calling thread.
 *   - host buffers are caller-owned, read (or written) during the call and never retained;
 *     *_device variants take device pointers valid on the context's GPU.
 *   - one context = one GPU = one HIP stream; handles are not thread-safe; calls that
 *     return data to the host are synchronous on return.
 *   - one process drives one GPU; multi-GPU = one process per GPU joined with
 *     pcr_comm_init (RCCL over xGMI), the scan sharded across ranks.
 *   - point clouds are float32 (N,3) row-major unless stated.  4x4 transforms are float64
 *     row-major.  The normal equations come back as out[29]:
 *       out[0..20]  upper triangle of H (6x6) row-major: 00 01 02 03 04 05 11 12 .. 55
 *       out[21..26] g,  out[27] e2,  out[28] number of correspondences that passed the gate
 */
#ifndef PCR_H
#define PCR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PCR_API __attribute__((visibility("default")))

/* Bumped whenever an existing entry point changes its argument layout or the size of an array it writes
 * (4: PCR_K_COUNT 5 -> 6, i.e. pcr_profile_read writes six entries).  A binding compares it with
 * pcr_abi_version() of the library it loaded before calling anything else.                                   */
#define PCR_ABI_VERSION 5

typedef int pcr_status;
enum {
    PCR_OK = 0,
    PCR_ERR_INVALID = -1,     /* bad argument */
    PCR_ERR_HIP = -2,         /* HIP runtime failure (message has the HIP error string) */
    PCR_ERR_NO_TARGET = -3,   /* target lacks what the requested kind needs (normals / icov) */
    PCR_ERR_COMM = -4,        /* RCCL failure or communicator misuse */
    PCR_ERR_SINGULAR = -5,    /* pcr_align: H singular (zero correspondences); quirk Q7 */
    PCR_ERR_NOMEM = -6,
    PCR_ERR_UNSUPPORTED = -7  /* an optional fast path is not available for this input (the target stays usable as it was):
                                 pcr_target_points_set_f64 on coordinates float32 cannot resolve */
};

/* registration kinds: which reference class's calc_H_g_e2 is evaluated */
enum {
    PCR_ICP = 0,      /* icp.py:24-57 */
    PCR_PLANE = 1,    /* plane_icp.py:30-69 */
    PCR_VPLANE = 2,   /* voxelized_plane_icp.py:23-64 */
    PCR_NDT = 3       /* ndt.py:24-57 */
};

/* compat flags */
enum {
    PCR_FLAG_ICP_RR_QUIRK = 1u,   /* reproduce icp.py:53-54: g[3:] = sum p x (R r) (quirk Q1); default */
    PCR_FLAG_NO_SCAN_SORT = 2u,   /* pcr_scan_create: keep the caller's point order on the device */
    PCR_FLAG_LOCAL_ONLY = 4u,     /* pcr_linearize / pcr_align: this rank's sums only, no all-reduce even when a
                                     communicator is attached to the context (the collective is a per-call decision) */
    PCR_FLAG_HOST_LOOP = 8u,      /* pcr_align: host-driven loop (one pcr_linearize + host solve per iteration)
                                     instead of the device-resident one; same arithmetic, same bits */
    PCR_FLAG_DEVICE_LOOP = 16u    /* pcr_align: device-resident loop even where the library would pick the host-driven
                                     one (small scans on one GPU, where it is ~8 % faster: 37.6 vs 41.0 us per iteration
                                     on a 100 k-point scan) */
};

typedef struct pcr_context pcr_context;
typedef struct pcr_target pcr_target;
typedef struct pcr_scan pcr_scan;
typedef struct pcr_group pcr_group;                 /* single-process multi-device: one context + host thread per member */
typedef struct pcr_group_target pcr_group_target;   /* the same target index on every member */
typedef struct pcr_group_scan pcr_group_scan;       /* a scan cut into one contiguous shard per member */

/* ---- library / context ------------------------------------------------------------ */
PCR_API const char *pcr_last_error(void);
PCR_API const char *pcr_version(void);
PCR_API int pcr_abi_version(void);
PCR_API pcr_status pcr_device_count(int *count);
PCR_API pcr_status pcr_context_create(int device, pcr_context **out);
PCR_API pcr_status pcr_context_destroy(pcr_context *ctx);
/* the context's HIP stream (hipStream_t as void*), for callers that order their own work */
PCR_API pcr_status pcr_context_stream(pcr_context *ctx, void **stream);
PCR_API pcr_status pcr_context_synchronize(pcr_context *ctx);
/* Destroying targets and scans does not return their device memory to the HIP allocator: the blocks go to a
 * per-context cache (hipFree synchronises the device, ~60 us each) and are handed out again to later targets /
 * scans of a similar size.  At most PCR_CACHE_LIMIT_MB (default 1024) of idle blocks are kept; they are invisible to
 * other allocators of the process (torch) until pcr_context_trim, which synchronises the stream and hipFree()s all
 * of them (released_bytes may be NULL), or an allocation failure inside the library, which does the same.        */
PCR_API pcr_status pcr_context_trim(pcr_context *ctx, uint64_t *released_bytes);

/* ---- multi-GPU: one process per GPU, RCCL all-reduce of the 29 doubles per iteration ----
 * No reference counterpart (the reference is single-process).  Rank 0 obtains an id with
 * pcr_comm_unique_id and distributes the 128 bytes out of band (bench.py uses
 * torch.distributed); every rank then calls pcr_comm_init.  After that pcr_linearize /
 * pcr_align return the SUM over all ranks' scan shards, unless the call carries
 * PCR_FLAG_LOCAL_ONLY (no collective is issued then; every rank of the communicator must
 * make the same choice for a given call).                                                */
PCR_API pcr_status pcr_comm_unique_id(void *id128);
PCR_API pcr_status pcr_comm_init(pcr_context *ctx, const void *id128, int nranks, int rank);
PCR_API pcr_status pcr_comm_destroy(pcr_context *ctx);
/* The same exchange WITHOUT a collective library in the loop (opt-in; RCCL stays the default): every rank exports a block of
 * slots (pcr_comm_p2p_export -> 64 bytes = a hipIpcMemHandle_t, distributed out of band like the RCCL id), maps its peers'
 * (pcr_comm_p2p_attach: handles = nranks x 64 bytes in rank order, at most 8 ranks), and a one-wave kernel between fold and
 * hand-off stores the 29 doubles + a sequence word into every rank's block, waits for everybody's words in its own and sums
 * the slots in rank order (bit-identical sums on every rank).  Exercised with two processes on one GPU; not yet on xGMI.
 * pcr_comm_p2p_failed reports whether an exchange gave up waiting for a peer (its sums are NaN then).                   */
PCR_API pcr_status pcr_comm_p2p_export(pcr_context *ctx, void *handle64);
PCR_API pcr_status pcr_comm_p2p_attach(pcr_context *ctx, const void *handles, int nranks, int rank);
PCR_API pcr_status pcr_comm_p2p_failed(pcr_context *ctx, int *failed);
/* 1: this context's slots are fine-grained device memory (coherent across devices while a kernel runs); 0: the coarse-grained
 * fallback, sound only between ranks that share ONE device -- ranks on different devices must agree on another transport
 * (distributed.py does).  PCR_P2P_ALLOW_COARSE=0 makes pcr_comm_p2p_export fail instead of falling back.
 * Since round 6 a timed-out exchange makes pcr_linearize / pcr_align of THAT rank return PCR_ERR_COMM (the device-resident
 * loop stops on it) instead of NaN sums with PCR_OK.                                                                    */
PCR_API pcr_status pcr_comm_p2p_finegrained(pcr_context *ctx, int *finegrained);

/* ---- single-process multi-device groups ("pcr_init(device_ids, n_dev)"; the reference is one process:
 * registration.py:28,71) ----
 * pcr_group_create: one context + one host thread per entry of device_ids (1..8 entries; an id may repeat -- [0, 0] are two
 * contexts on one GPU, which is how a one-GPU box exercises N > 1).  Targets are built once per member (the index is
 * replicated), pcr_group_scan_create cuts the scan into contiguous, balanced shards (member i: points [lo_i, hi_i), the
 * bounds of distributed.shard_bounds), and pcr_group_linearize / pcr_group_align run pcr_linearize / pcr_align on every
 * member at once with the members' 29 sums exchanged through the peer-to-peer kernel above on in-process peer pointers
 * (hipDeviceEnablePeerAccess; no IPC, no RCCL, no torch): every member takes the same Gauss-Newton step on bit-identical
 * sums, and the call returns member 0's copy -- what the SPMD run with the same sharding returns, bit for bit.
 * Calls on one group are serialised by the caller (like every other handle); errors: the first failing member's status,
 * pcr_last_error() names the member.  PCR_FLAG_LOCAL_ONLY is ignored (a group call is the sum over its members).          */
PCR_API pcr_status pcr_group_create(const int *device_ids, int n, pcr_group **out);
PCR_API pcr_status pcr_group_destroy(pcr_group *g);
PCR_API pcr_status pcr_group_size(pcr_group *g, int *n);
PCR_API pcr_status pcr_group_context(pcr_group *g, int i, pcr_context **ctx);            /* borrowed */
PCR_API pcr_status pcr_group_target_points_create(pcr_group *g, const float *xyz, int64_t n, const float *normals_or_null,
                                                  float cell_hint, pcr_group_target **out);
PCR_API pcr_status pcr_group_target_voxels_create(pcr_group *g, const void *xyz, int xyz_is_f64, int64_t n, double voxel_size,
                                                  int min_points, pcr_group_target **out);
PCR_API pcr_status pcr_group_target_estimate_normals(pcr_group_target *gt, int k, int compat, float *normals_out_or_null);
PCR_API pcr_status pcr_group_target_set_normals(pcr_group_target *gt, const float *normals);
PCR_API pcr_status pcr_group_target_points_set_f64(pcr_group_target *gt, const double *xyz64);
PCR_API pcr_status pcr_group_target_member(pcr_group_target *gt, int i, pcr_target **t);  /* borrowed */
PCR_API pcr_status pcr_group_target_destroy(pcr_group_target *gt);
PCR_API pcr_status pcr_group_scan_create(pcr_group *g, const float *xyz, int64_t n, unsigned flags, pcr_group_scan **out);
PCR_API pcr_status pcr_group_scan_size(pcr_group_scan *gs, int64_t *n);
PCR_API pcr_status pcr_group_scan_destroy(pcr_group_scan *gs);
PCR_API pcr_status pcr_group_linearize(pcr_group_target *gt, pcr_group_scan *gs, int kind, const double T[16], double max_dist,
                                       unsigned flags, double out[29]);
PCR_API pcr_status pcr_group_align(pcr_group_target *gt, pcr_group_scan *gs, int kind, const double T_init[16], int max_iter,
                                   double tol, double max_dist, unsigned flags, double T_out[16], int *iterations,
                                   double *trace_or_null);

/* ---- targets ------------------------------------------------------------------------
 * pcr_target_points_create replaces ICP.set_target (icp.py:17-22) and the KD-tree half of
 * PlaneICP.set_target (plane_icp.py:19-22): float32 copy of the cloud + an exact-NN index
 * (a dense cell grid in HBM instead of pykdtree's KD-tree, kdtree.py:18-21).
 * normals (N,3) float32 may be NULL; cell_hint = 0 picks the cell size automatically.     */
PCR_API pcr_status pcr_target_points_create(pcr_context *ctx, const float *xyz, int64_t n,
                                            const float *normals_or_null, float cell_hint,
                                            pcr_target **out);
PCR_API pcr_status pcr_target_points_create_device(pcr_context *ctx, const float *d_xyz, int64_t n,
                                                   const float *d_normals_or_null, float cell_hint,
                                                   pcr_target **out);
/* caller-supplied normals, PlaneICP.set_target(target, kdree, norm) (plane_icp.py:25-27)  */
PCR_API pcr_status pcr_target_set_normals(pcr_target *t, const float *normals);
/* Quirk Q6 -- PlaneICP.set_target builds its KD-tree on the ORIGINAL array (plane_icp.py:22) and gathers from the float32
 * copy (plane_icp.py:20,44): a float64 target is SEARCHED in float64 (queries up-cast, float64 distances, float64 gate,
 * plane_icp.py:40-41); so is KDTree(float64 data).query (kdtree.py:18-21).  xyz64 = the (N,3) float64 array whose
 * float32 rounding the target was created from; PCR_PLANE passes and pcr_nn_query_f64 then return the float64 search's
 * neighbour (float32 filter search over the index + float64 check, float64 box search for what that cannot separate);
 * PCR_ICP keeps the float32 search (icp.py:19-20 builds its tree on the float32 copy).  +32 bytes per point.          */
PCR_API pcr_status pcr_target_points_set_f64(pcr_target *t, const double *xyz64);
/* k-NN PCA normals, estimate_norm_with_tree (estimate_normals.py:27-87).  compat != 0 keeps
 * the reference's float32 single-pass covariance; normals_out (N,3) may be NULL.           */
PCR_API pcr_status pcr_target_estimate_normals(pcr_target *t, int k, int compat, float *normals_out);
PCR_API pcr_status pcr_target_get_normals(pcr_target *t, float *normals_out);

/* Voxel targets, VPlaneICP.set_target / NDT.set_target (voxelized_plane_icp.py:18-21,
 * ndt.py:18-22) = VoxelGrid.set_points + calc_icov (voxel.py:104-165, 69-102) built on the
 * GPU: hash keys (voxel.py:12-21), group, mean, two-pass covariance, min_points filter,
 * smallest-eigenvector normal, closed-form inverse covariance, and an exact
 * nearest-CENTROID index (voxel.py:165,171-179).  xyz_is_f64 selects the dtype the keys
 * are computed in (float32 clouds divide in float32, as NumPy does).                      */
PCR_API pcr_status pcr_target_voxels_create(pcr_context *ctx, const void *xyz, int xyz_is_f64, int64_t n,
                                            double voxel_size, int min_points, pcr_target **out);
/* the same from precomputed statistics (mean, norm: (n_v,3) float64; icov: (n_v,3,3) float64,
 * norm / icov may be NULL): lets tests inject the oracle's voxels for kernel-only parity.   */
PCR_API pcr_status pcr_target_voxels_create_from_stats(pcr_context *ctx, const double *mean,
                                                       const double *norm_or_null, const double *icov_or_null,
                                                       int64_t n_v, double voxel_size, pcr_target **out);
/* read the voxel statistics back (any pointer may be NULL); *n_v is always written          */
PCR_API pcr_status pcr_target_voxels_get(pcr_target *t, int64_t *n_v, double *mean, double *cov,
                                         double *norm, double *icov, int64_t *counts, int64_t *keys);
PCR_API pcr_status pcr_target_size(pcr_target *t, int64_t *n);
PCR_API pcr_status pcr_target_destroy(pcr_target *t);

/* ---- scan ---------------------------------------------------------------------------
 * Registration.align casts the scan to float32 once (registration.py:83) and reuses it for
 * every iteration: uploaded once, Morton-sorted on the device for gather coherence (the
 * sums do not depend on point order beyond rounding).                                      */
PCR_API pcr_status pcr_scan_create(pcr_context *ctx, const float *xyz, int64_t n, unsigned flags, pcr_scan **out);
PCR_API pcr_status pcr_scan_create_device(pcr_context *ctx, const float *d_xyz, int64_t n, unsigned flags,
                                          pcr_scan **out);
PCR_API pcr_status pcr_scan_size(pcr_scan *s, int64_t *n);
PCR_API pcr_status pcr_scan_destroy(pcr_scan *s);
/* Test / diagnostic seam: the correspondences the last search + reduce pass over this scan left in HBM -- one
 * word per scan point IN THE SCAN'S DEVICE ORDER (Morton-sorted unless PCR_FLAG_NO_SCAN_SORT): the index of the
 * matched record in the target's cell-sorted arrays, 0xffffffff = no correspondence.  Lets a test compare two
 * search kernels match by match (KDTree.query's idx, kdtree.py:18-21 / voxel.py:171-179); fails when the scan
 * has only run the fused small-scan kernel, which keeps its matches in registers.                            */
PCR_API pcr_status pcr_scan_read_matches(pcr_scan *s, uint32_t *out);

/* calc_H_g_e2(cur_T, source) takes the scan as an array on every call and is pure in it (registration.py:55-68):
 * the drop-in class keeps the device copy of the last scan and re-uploads when the CONTENT of the caller's array
 * changed.  pcr_hash64 is the content hash it uses: 64-bit, non-cryptographic, multi-threaded (12.7 MB in ~0.1 ms
 * on the GPU box's host; the value does not depend on the number of threads).                               */
PCR_API pcr_status pcr_hash64(const void *data, uint64_t nbytes, uint64_t *out);
/* LZF, the codec of PCD "DATA binary_compressed" (the real data/B-01.pcd of benchmark/test_data.py:11,24 may be stored that
 * way): decompress in_len bytes into out (capacity out_len; *written = bytes produced; PCR_ERR_INVALID on a corrupt or
 * oversized stream), and a greedy compressor for writers and tests (out_cap >= in_len + in_len / 32 + 8 always suffices).  Host
 * only, no GPU involved.                                                                                                  */
PCR_API pcr_status pcr_lzf_decompress(const void *in, uint64_t in_len, void *out, uint64_t out_len, uint64_t *written);
PCR_API pcr_status pcr_lzf_compress(const void *in, uint64_t in_len, void *out, uint64_t out_cap, uint64_t *written);
/* CPUs the process may actually use: affinity mask capped by the cgroup CPU-bandwidth quota (cpu.max).  The hash
 * pool above is sized from it; a host application should size ITS thread pools (OpenMP, BLAS) the same way -- on a
 * 256-CPU box inside a 16-CPU container, pools sized by the visible CPUs get every thread of the process parked for
 * the rest of a 100 ms scheduler period, the calling thread of pcr_linearize included.          */
PCR_API int pcr_usable_cpus(void);

/* ---- the hot path ---------------------------------------------------------------------
 * One calc_H_g_e2: transform (math_tools.py:111-113) -> exact 1-NN (kdtree.py:18-21 /
 * voxel.py:171-179) -> gate dist < max_dist -> residual + Jacobian -> 6x6 normal equations
 * (icp.py:24-57, plane_icp.py:30-69, voxelized_plane_icp.py:23-64, ndt.py:24-57), summed
 * over all ranks when a communicator is attached.  The search is bounded by max_dist, which
 * is exact for these sums (anything farther is gated out anyway).                          */
PCR_API pcr_status pcr_linearize(pcr_target *t, pcr_scan *s, int kind, const double T[16],
                                 double max_dist, unsigned flags, double out[29]);

/* Registration.align (registration.py:71-113) run entirely behind the boundary: up to
 * max_iter x { pcr_linearize, dx = -solve(H, g), stop if |dx| < tol (before the update,
 * quirk Q4), T <- plus(T, dx) (math_tools.py:101-108, first-order expSO3 branch, quirk Q3) }.
 * The loop is device-resident (large scans, and always with a communicator): the pose stays in HBM, the 6x6 solve and the update run in a one-wave
 * kernel (k_gn_update) behind the reduce kernel, iterations are enqueued back to back and the host reads one
 * result (PCR_FLAG_HOST_LOOP selects the host-driven form of the same arithmetic).
 * trace_or_null receives up to max_iter rows of 16 (T before the step) + 29 doubles.
 * Returns PCR_ERR_SINGULAR where numpy.linalg.solve would raise LinAlgError (quirk Q7).     */
PCR_API pcr_status pcr_align(pcr_target *t, pcr_scan *s, int kind, const double T_init[16],
                             int max_iter, double tol, double max_dist, unsigned flags,
                             double T_out[16], int *iterations, double *trace_or_null);

/* ---- fine seam: KDTree(data).query(points, k) (kdtree.py:18-65) ------------------------
 * dist is Euclidean (not squared).  For point targets dist is float32 and idx indexes the
 * array given to pcr_target_points_create; for voxel targets use the f64 variant (idx =
 * kept-voxel index).  r_max <= 0 or inf = unbounded (as the reference); with a bound,
 * queries with no neighbour inside r_max get idx = -1, dist = inf.                         */
PCR_API pcr_status pcr_nn_query(pcr_target *t, const float *q, int64_t m, float r_max, float *dist, int64_t *idx);
PCR_API pcr_status pcr_nn_query_f64(pcr_target *t, const float *q, int64_t m, double r_max, double *dist, int64_t *idx);
PCR_API pcr_status pcr_knn_query(pcr_target *t, const float *q, int64_t m, int k, float *dist, int64_t *idx);

/* ---- instrumentation ------------------------------------------------------------------
 * With profiling on, every launch of a hot-path kernel is bracketed by HIP events on the
 * context's stream; pcr_profile_read drains them (synchronises) and reports per-kernel
 * launch count and total milliseconds since the last reset.  on = n > 1 brackets only every
 * n-th pass (an event pair costs a few microseconds of stream time: sampling keeps a timed
 * region honest); on = 1 every pass; 0 off.  pcr_align's device-resident loop keeps up to two
 * iterations queued beyond the one that converges: those launches return at once but are
 * bracketed like the others, so per-kernel averages taken over an align include them.        */
enum { PCR_K_LINEARIZE = 0, PCR_K_FINALIZE = 1, PCR_K_NN = 2, PCR_K_REDUCE = 3, PCR_K_ALLREDUCE = 4, PCR_K_CERTIFY = 5, PCR_K_COUNT = 6 };
PCR_API pcr_status pcr_profile_enable(pcr_context *ctx, int on);
PCR_API pcr_status pcr_profile_reset(pcr_context *ctx);
PCR_API pcr_status pcr_profile_read(pcr_context *ctx, int64_t launches[PCR_K_COUNT], double total_ms[PCR_K_COUNT]);
/* the same with the caller's array capacity stated (at most `capacity` entries are written; *count = PCR_K_COUNT of
 * the library): what a binding compiled against an older header should call                                       */
PCR_API pcr_status pcr_profile_read_n(pcr_context *ctx, int capacity, int64_t *launches, double *total_ms, int *count);
/* grid geometry of a target's NN index: cell size, dims, occupied cells, points (or voxels) */
PCR_API pcr_status pcr_target_index_info(pcr_target *t, double *cell, int64_t dims[3], int64_t *occupied, int64_t *n);
/* How unevenly the points fill the index (round 6): the largest and the 99th-percentile population of an OCCUPIED cell, and
 * whether the target was built as "heavy" (some cells hold far more points than the average -- a LiDAR sweep's ring lines,
 * density ~ 1/r^2): its cell edge then stops at PCR_CELLS_PER_POINT (default 8) grid cells per point instead of shrinking to
 * ~5 points per occupied cell, the points of a cell are Morton-sorted, and every range of more than 24 records is searched
 * through boxes over 64 and 8 consecutive records (csrc/nn_device.h: nn_scan_range_lb).  Any pointer may be NULL.         */
PCR_API pcr_status pcr_target_index_population(pcr_target *t, int64_t *pop_max, int64_t *pop_p99, int *heavy);
/* point targets: margin (metres) and total records of the extended per-cell lists ring 0 searches (a cell's
 * own points plus the neighbours' points within the margin of the shared face; PCR_HALO sets the margin as a
 * fraction of the cell edge, default 0.1, 0 = none).  Voxel targets: the same of the float32 filter index over
 * the rounded centroids (both 0: the target has no filter -- coordinates too large -- and searches in float64) */
PCR_API pcr_status pcr_target_index_halo(pcr_target *t, double *halo, int64_t *records);
/* point targets: the same of the second, deeper set of lists (margin 0.25 x cell), which the library builds once a target
 * has served a dozen search + reduce passes and which serves the passes of a Gauss-Newton run whose scan still moves by
 * more than 0.12 cell per pass; both 0 while it does not exist                                                         */
PCR_API pcr_status pcr_target_index_halo2(pcr_target *t, double *halo, int64_t *records);
/* voxel targets: the bound (metres) on how far rounding to float32 moved any centroid of the filter index -- what the
 * float32 filter search of the centroid search (voxel.py:171-179) subtracts from its lower bounds before the float64
 * check; 1.75 x 2^-24 x the largest coordinate magnitude of the centroid grid.  0 = no filter index (not built yet:
 * the first search + reduce pass builds it; or refused: band > 1 % of the index cell)                               */
PCR_API pcr_status pcr_target_filter_band(pcr_target *t, double *band);
/* work counters of the NN search for one pose (point targets): out[0..3] = rings entered, row
 * segments loaded, rows pruned by arithmetic, candidates tested, summed over queries; out[4..7] = the
 * same with each wave's maximum charged to all 64 lanes (the cost under divergence); out[8..10] =
 * wave wall-clock cycles summed over waves: prologue (load, transform, cell), ring 0, outer rings   */
PCR_API pcr_status pcr_nn_counters(pcr_target *t, pcr_scan *s, const double T[16], double max_dist, double out[11]);
/* select the hot-path variant: 2 (default) = chosen per launch by the size of the scan: 1 = NN kernel
 * writing correspondences to HBM followed by a reduce kernel (faster for large scans: twice the
 * occupancy), 0 = one fused transform+NN+reduce kernel (faster for small, latency-bound scans: 100 k
 * points 49 vs 59 us per pass); either way the last blocks fold the partial sums inside the kernel   */
PCR_API pcr_status pcr_set_variant(pcr_context *ctx, int variant);
PCR_API pcr_status pcr_get_variant(pcr_context *ctx, int *variant);
/* NN search kernel of variant 1: 0 = per-lane ring search (shipped; plain passes over a voxel target run a float32 filter
 * search over the rounded centroids and check its winner in float64 -- results identical to the float64 search),
 * 2 = wave-cooperative LDS-staged search, 3 = as 0 with the centroid search in float64 throughout (A/B, tests),
 * 4 = wave-cooperative search with an MFMA distance filter (round 5); 2 and 4: developer build only                */
PCR_API pcr_status pcr_set_nn_mode(pcr_context *ctx, int mode);
/* Certified reuse of the previous pass' matches (no reference counterpart: Registration.align,
 * registration.py:89-111, searches afresh every iteration).  When consecutive passes over one scan and target
 * differ by a small pose change, a pass first proves -- per point, by the triangle inequality on a bound the
 * previous search recorded -- that the old match is still the exact nearest neighbour, and searches only the
 * points where the proof fails.  The correspondences, and therefore all 29 sums, are bit-identical to a full
 * search; only the time changes.  mode: 0 = off (the DEFAULT since round 5: at the reference's tol = 1e-3 a Gauss-Newton
 * run stops as soon as its steps reach millimetres, the automatic policy never engaged on a BASELINE config, and the state
 * costs 4 bytes per scan point -- opt in for tight tolerances / re-evaluation in place, where it pays 1.2-2.6x), 1 =
 * automatic (tried when the scan's typical displacement since the last pass is below tau x cell size of the target's
 * index), 2 = always.  mu = how far beyond its match a tracking search looks, x cell size.  tau / mu <= 0 keep the
 * current value.  PCR_REUSE in the environment sets the mode of new contexts.                                          */
PCR_API pcr_status pcr_set_reuse(pcr_context *ctx, int mode, double tau, double mu);
PCR_API pcr_status pcr_get_reuse(pcr_context *ctx, int *mode, double *tau, double *mu);
/* out[0..2] = passes over this scan by search mode (full, tracking, certify + list); out[3] / out[4] = points
 * the list passes had to search / points they covered; out[5..7] = mode, searched points (-1 = not read
 * back) and typical displacement (m) of the last pass                                                      */
PCR_API pcr_status pcr_scan_reuse_stats(pcr_scan *s, double out[8]);
/* variant 1: 1 (default) = the reduce kernel folds the per-block partial sums itself
 * (k_reduce_finalize); 0 = separate fold kernel                                             */
PCR_API pcr_status pcr_set_fuse_finalize(pcr_context *ctx, int on);
/* 1 when the library carries the developer / A-B kernels (unfused folds, the wave-cooperative LDS-staged search, the work
 * counters: csrc/kernels_dev.hip) -- libpcr_hip_dev.so, built by `make DEV=1`; the shipped libpcr_hip.so returns 0 and
 * refuses pcr_set_nn_mode(2), pcr_set_fuse_finalize(0) and pcr_nn_counters with PCR_ERR_INVALID.                      */
PCR_API int pcr_has_dev_kernels(void);
PCR_API pcr_status pcr_get_pipeline(pcr_context *ctx, int *variant, int *fuse_finalize, int *nn_mode);

#ifdef __cplusplus
}
#endif
#endif /* PCR_H */
