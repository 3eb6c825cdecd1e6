// Internal declarations shared by the translation units of libpcr_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <mutex>
#include <unordered_map>
#include <utility>
#include <vector>

#include "pcr.h"

void pcr_set_error(const char *fmt, ...);

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            pcr_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return PCR_ERR_HIP;                                                                \
        }                                                                                      \
    } while (0)

#define PCR_TRY(expr)                      \
    do {                                   \
        pcr_status s_ = (expr);            \
        if (s_ != PCR_OK) return s_;       \
    } while (0)

#define PCR_REQUIRE(cond, msg)                      \
    do {                                            \
        if (!(cond)) {                              \
            pcr_set_error("invalid argument: %s", msg); \
            return PCR_ERR_INVALID;                 \
        }                                           \
    } while (0)

// Dense cell grid geometry.  Real = float for point targets, double for voxel centroids.
template <typename Real>
struct Geom {
    Real ox, oy, oz;   // min corner
    Real h, inv_h;     // cell edge and its reciprocal
    Real slack;        // conservative margin on every pruning bound (rounding of cell assignment)
    int nx, ny, nz;
    // cell_start[c] = (gap << 28) | first point of cell c, where gap = min(15, Chebyshev distance in
    // cells to the nearest occupied cell): rings closer than `gap` around c are empty and skipped.
    // cs_mask = 0x0fffffff when the gap field is present (targets below 2^28 points), else ~0.
    uint32_t cs_mask;
    // seed[c] (cells with gap > 0 only matter): index of a point in the nearest occupied cell; its
    // distance initialises the search bound of a query that lands in empty space
    const uint32_t *seed;
    // halo (point targets, 0 = none): cs_h / pts_h = per-cell EXTENDED lists, a cell's own points plus the
    // points of its 26 neighbours within `halo` of the shared face / edge / corner; cs_h carries the same
    // gap bits as cell_start; j_h[e] = cell-sorted index of the point copied to pts_h[e].  Ring 0 scans the
    // extended list and certifies everything within fmin + halo.
    Real halo;
    const uint32_t *cs_h;
    const void *pts_h;
    const uint32_t *j_h;
    // row-occupancy bitmap (nullptr = none): word ((z * nyw + (y >> 6)) * nxb + (x >> 4)), bit y & 63 = row (y, z) has a
    // point in cells [16 (x >> 4), 16 (x >> 4) + 16): lets the wide rings of a far query skip empty rows 64 at a time
    const unsigned long long *rowocc;
    int nyw, nxb;
    // row-block boxes (float32 point targets, nullptr = none; round 6): record ((z * ny + y) * nxr + (x >> 3)) describes the
    // points in cells [8 (x >> 3), 8 (x >> 3) + 8) of row (y, z): .x = occupancy of the 8 cells (byte 0) | x_lo (byte 1) |
    // x_hi (byte 2) | y_lo (byte 3), .y = y_hi (byte 0) | z_lo (byte 1) | z_hi (byte 2).  x bounds in units of h / 32 from the
    // block's first cell, y / z bounds in units of h / 256 from the row's cell; a bound byte b stands for [b, b + 1).  The
    // far search (nn_device.h: nn_rings_box) prunes a row segment by its distance to this TIGHT box instead of to the cubes
    // of its cells: clouds are surfaces, and a surface fills a small part of the cells it crosses.
    const uint2 *rbox;
    int nxr;
    // leaf / group boxes (float32 point targets with HEAVY cells, nullptr = none; round 6): lbox[L] = the box of records
    // [8 L, 8 L + 8) of the cell-sorted array (one 16-byte record: nn_device.h box_d2), gbox[G] the same for records [64 G, 64 G + 64) -- fixed blocks of
    // the ARRAY, so a block may straddle cells (its box is then looser, never wrong).  The points of a cell are sorted along a
    // Morton curve inside the cell, so a block is a compact patch.  A range of more than PCR_LB_MIN records is scanned box by
    // box (nn_scan_range_lb): a cell of a LiDAR sweep's inner rings holds hundreds of points where the average cell holds five.
    // lbox_h / gbox_h: the same over the extended lists (pts_h).
    const float4 *lbox, *gbox, *lbox_h, *gbox_h;
};
#define PCR_LB_MIN 24            // ranges up to this many records are scanned plainly
#define PCR_RB_LOG 3             // cells per row block = 8
#ifndef PCR_HALO2_FRAC
#define PCR_HALO2_FRAC 0.25      // margin of the deeper list set, x cell
#endif
#define PCR_HALO2_AFTER 12       // passes a point target serves before the deeper set is built (~0.3 ms: repaid after ~60 passes)
#ifndef PCR_HALO2_MOVE
#define PCR_HALO2_MOVE 0.12      // the deeper set serves passes whose scan moved by at least this x cell since the previous pass
#endif
#define PCR_GAP_SHIFT 28
#define PCR_GAP_MAX 15

// points of a grid, cell-sorted: xyz + original index bit-cast into w
typedef float4 PtF;
typedef double4 PtD;
// PlaneICP gather record, cell-sorted, 32-byte aligned: the matched point AND its normal arrive in
// ONE 32-byte sector (two separate float4 arrays cost two 64-byte lines per correspondence once the
// target outgrows the caches: 94 vs 40 algorithmic bytes per query at 1e8 points)
struct __attribute__((aligned(32))) PtN {
    float x, y, z;
    uint32_t orig;
    float nx, ny, nz;
    uint32_t pad;
};

// Device-resident state of the Gauss-Newton loop (pcr_align): the pose every kernel of an iteration
// reads, rewritten by k_gn_update (solve + boxplus on the device, one wave, its own launch).
struct PoseDev {
    double T[16];          // current pose, row-major
    double R[9];           // = T[:3,:3]
    float r32[9], t32[3];  // float32 copy used to transform the scan (math_tools.py:111-113)
    int iter;              // passes completed
    int done;              // 0 running, 1 converged, 2 singular, 3 max_iter reached
    int tile_local;        // hand-out policy of the next search, decided by k_gn_update from the size of its step
    int halo_deep;         // the next search reads the deeper set of extended lists (the step was large), same decision point
};
#define PCR_LOOP_RUNNING 0
#define PCR_LOOP_CONVERGED 1
#define PCR_LOOP_SINGULAR 2
#define PCR_LOOP_MAXITER 3
#define PCR_LOOP_COMMFAIL 4   // an exchange of this rank gave up waiting for a peer (comm.hip: k_p2p_allreduce)

// ---- temporaries: a per-context cache of device blocks ------------------------------------------
// hipFree synchronises the device and costs ~60 us; a target / scan / voxel build used to issue 10-40 of
// them (2.4 of the 3.8 ms of a 1.06 M-point voxel build).  Temporaries (DevBuf) created while a context is
// "current" on the calling thread (CtxScope, set by the API entry points) are returned to that context's
// cache instead and handed out again to later requests of a similar size.  Everything a context does runs
// on its ONE stream, so reusing a block is ordered behind its previous user.  Blocks are whole hipMalloc
// allocations: release() can still hand one over to a persistent owner that hipFree()s it later.
struct pcr_context;
void *pcr_cache_get(pcr_context *ctx, size_t bytes, size_t *cap_out, bool tight = false);     // nullptr: nothing suitable cached
void pcr_cache_put(pcr_context *ctx, void *p, size_t cap);
void pcr_cache_clear(pcr_context *ctx);
extern thread_local pcr_context *pcr_tls_ctx;
// hipMalloc; on failure the current context's idle blocks (up to 1 GiB) are released and the call is retried once
hipError_t pcr_malloc_retry(void **p, size_t bytes);
// Persistent blocks (what a target keeps): taken from the current context's cache when a block fits TIGHTLY (at most
// 1/8 + 256 KiB larger), registered with their capacity in that context (pcr_context::owned), and handed back to its cache
// by pcr_persist_free -- set_target over an old target used to pay ~17 hipFree + ~17 hipMalloc calls (~60 us each).
hipError_t pcr_persist_alloc(void **p, size_t bytes);
void pcr_persist_free(pcr_context *ctx, void *p);          // unregistered pointers are hipFree()d; nullptr: no-op
struct CtxScope {
    pcr_context *prev;
    explicit CtxScope(pcr_context *ctx) : prev(pcr_tls_ctx) { pcr_tls_ctx = ctx; }
    ~CtxScope() { pcr_tls_ctx = prev; }
};

// Device allocation released on scope exit unless handed over with release(): every early return
// of the HIP_TRY / PCR_TRY macros leaves no temporaries behind.
template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;                 // bytes of the underlying block
    pcr_context *owner = nullptr;   // context whose cache the block goes back to (nullptr: plain hipFree)
    bool persist = false;           // block registered in owner->owned (alloc_exact)
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { reset(); }
    hipError_t alloc(size_t count) { return alloc_bytes(sizeof(T) * (count ? count : 1)); }
    // tightly sized, registered as a persistent block: for buffers that will be release()d to a persistent owner,
    // which frees them with pcr_persist_free
    hipError_t alloc_exact(size_t count) {
        reset();
        owner = pcr_tls_ctx;
        persist = true;
        cap = sizeof(T) * (count ? count : 1);
        const hipError_t e = pcr_persist_alloc((void **)&p, cap);
        if (e != hipSuccess) { p = nullptr; cap = 0; }
        return e;
    }
    hipError_t alloc_bytes(size_t bytes) {
        reset();
        if (bytes == 0) bytes = 16;
        owner = pcr_tls_ctx;
        persist = false;
        if (owner) {
            p = (T *)pcr_cache_get(owner, bytes, &cap);
            if (p) return hipSuccess;
            bytes = (bytes + 4095) & ~(size_t)4095;
        }
        cap = bytes;
        const hipError_t e = pcr_malloc_retry((void **)&p, bytes);
        if (e != hipSuccess) { p = nullptr; cap = 0; }
        return e;
    }
    void reset() {
        if (p) {
            if (persist) pcr_persist_free(owner, p);
            else if (owner) pcr_cache_put(owner, p, cap);
            else (void)hipFree(p);
        }
        p = nullptr; cap = 0;
    }
    T *release() {
        T *q = p;
        p = nullptr; cap = 0;
        return q;
    }
    operator T *() const { return p; }
};

struct ProfEvent {
    int kernel;
    hipEvent_t start, stop;
};

struct pcr_context {
    int device = 0;
    int num_cu = 256;
    hipStream_t stream = nullptr;
    // reduction scratch
    double *d_partials = nullptr;   // [max_blocks][32]
    int max_blocks = 0;
    double *d_out = nullptr;        // 32 doubles (29 used)
    double *h_out = nullptr;        // pinned + mapped: 32 doubles, then the completion sequence number
    double *h_out_dev = nullptr;    // device-side address of h_out
    uint32_t seq = 0;
    uint32_t passes_since_query = 0;   // see retire_completed (kernels.hip)
    bool stall_debug = false;          // PCR_STALL_DEBUG: report where a pass slower than 1 ms spent its time
    int retire_period = 0;             // hipStreamQuery every n-th pass (PCR_RETIRE_PERIOD; 0 = never, the default)
    // device-resident Gauss-Newton loop: pose in HBM, per-iteration trace rows (16 + 29 doubles),
    // progress words in pinned host memory ([0] iter, [1] done, then 16 doubles of pose)
    PoseDev *d_pose = nullptr;
    double *d_trace = nullptr;
    int trace_cap = 0;
    int variant = 0;
    int nn_mode = 0;             // 0 per-lane search; 2 wave-cooperative, LDS-staged (developer builds); 3 = 0 without the float32 filter of the
                                 // centroid search; 4 = wave-cooperative with an MFMA distance filter (developer builds, round 5)
    // certified reuse of the previous pass' matches (see kernels.hip: choose_nn_mode)
    double local_frac = 0.35;    // block-local tile hand-out when the scan moved less than this x cell size (PCR_LOCAL_FRAC)
    double voxel_cell_mult = 2.0; // PCR_VOXEL_CELL_MULT: centroid grid cell edge in voxels
    int vox_filter = 1;          // PCR_VOX_FILTER=0: plain passes search the centroids in float64 (no float32 filter)
    int filter_after = 8;        // PCR_FILTER_AFTER: fused small-scan passes a voxel target serves before its filter index is built
    int vox_occ = -1;            // PCR_VOX_OCC (developer): force the centroid search's row-bitmap variant on / off; -1 = by gate / cell ratio
    int tile_local = -1;         // PCR_TILE_LOCAL (developer): force the hand-out policy of k_nn_scan; -1 = automatic
    int reuse = 0;               // 0 off (default since round 5: the automatic policy never engaged on a BASELINE config at tol 1e-3, and the
                                 // state costs 4 bytes per scan point), 1 automatic, 2 forced (track + list whenever the state allows: tests)
    double reuse_tau = 0.0125;   // try it when the scan's typical motion since the last pass is below tau x cell size (a quarter of mu)
    double reuse_mu = 0.05;      // margin of the tracking search, x cell size
    uint64_t next_serial = 1;    // targets get unique serial numbers (validity of a scan's previous matches)
    bool fuse_finalize = true;   // k_reduce_finalize (PCR_FUSE_FINALIZE=0: k_reduce + k_finalize)
    uint32_t *d_tile_ctr = nullptr;     // per-XCD dynamic tile counters of k_nn_scan
    int nn_blocks_per_cu[5] = {4, 4, 4, 4, 2};   // resident 256-thread blocks per CU of k_nn_scan<0/1>, k_nn_coop, k_nn_filter, k_nn_mfma
    int nn_blocks_rb = 4;               // ... of k_nn_scan<0, ., ., FULL, RB = 1> (row-block boxes)
    int nn_blocks_lb = 4;               // ... of k_nn_scan<0, ., ., FULL, RB = 2> (leaf / group boxes)
    int nn_blocks_ps = 4;               // ... of k_scan_reduce (phase-split search + reduce, round 6)
    int phase_split = 0;                // PCR_PHASE_SPLIT=1 (opt-in; measured slower, profiles/r06_phase_split_null.txt): mid-size scans over point targets run k_scan_reduce instead of k_nn_scan + k_reduce_finalize
    uint32_t filter_stamp = 0;          // stamp of the last k_nn_filter pass (k_nn_fix)
    // profiling
    bool prof_on = false;
    int prof_period = 1;        // events around every prof_period-th pass (1 = every pass)
    uint64_t prof_pass = 0;     // passes enqueued since profiling was switched on
    bool prof_this_pass = false;
    std::vector<ProfEvent> prof_events;
    std::vector<ProfEvent> prof_free;
    int64_t prof_launches[PCR_K_COUNT] = {0};
    double prof_ms[PCR_K_COUNT] = {0};
    // RCCL
    void *comm = nullptr;        // ncclComm_t (comm_kind 0) or the peer-to-peer state (comm_kind 1, comm.hip: P2PState)
    int comm_kind = 0;
    int nranks = 1, rank = 0;
    // cache of free device blocks (temporaries of the build paths): (capacity, pointer), total bytes.
    // cache / cache_bytes / owned are guarded by cache_mu: a Scan / Target finalizer on one host thread (ctypes drops the
    // GIL) may hand blocks back while another thread's pass or set_target takes blocks out.  Everything else
    // of a context stays single-threaded by contract (include/pcr.h).
    std::recursive_mutex cache_mu;
    size_t cache_limit = (size_t)1 << 30;        // idle bytes kept at most (PCR_CACHE_LIMIT_MB; pcr_context_trim empties it)
    std::vector<std::pair<size_t, void *>> cache;
    size_t cache_bytes = 0;
    std::unordered_map<void *, size_t> owned;    // persistent blocks handed out by pcr_persist_alloc: capacity
};

struct pcr_target {
    pcr_context *ctx = nullptr;
    int is_voxel = 0;
    int64_t n = 0;           // points or kept voxels
    int64_t occupied = 0;    // occupied cells of the NN grid
    uint32_t *cell_start = nullptr;
    uint32_t *cell_seed = nullptr;
    unsigned long long *rowocc = nullptr;   // row-occupancy bitmap of the grid (Geom::rowocc)
    uint2 *rbox = nullptr;                  // row-block boxes of a point target (Geom::rbox)
    float4 *lbox = nullptr, *gbox = nullptr, *lbox_h = nullptr, *gbox_h = nullptr, *lbox_h2 = nullptr, *gbox_h2 = nullptr;   // Geom::lbox ...
    int64_t pop_max = 0, pop_p99 = 0;       // largest / 99th-percentile population of an occupied cell (pcr_target_index_population)
    bool heavy = false;                     // some cells hold far more points than the average: boxes built, box-aware search
    uint32_t *cs_h = nullptr;        // extended (halo) lists of point targets
    PtF *pts_h = nullptr;
    uint32_t *j_h = nullptr;
    int64_t n_h = 0;                 // records in pts_h
    // a second, deeper set of the same lists (halo PCR_HALO2_FRAC x cell), built lazily once the target has served
    // PCR_HALO2_AFTER search + reduce passes; a pass reads it while the scan still moves by more than PCR_HALO2_MOVE x cell
    // per pass (measured per pose on plane_b01, halo 0.1 / 0.25: 232 / 232, 180 / 180, 105 / 89, 49 / 40, 32 / 39 us)
    uint32_t *cs_h2 = nullptr;
    PtF *pts_h2 = nullptr;
    uint32_t *j_h2 = nullptr;
    int64_t n_h2 = 0;
    float halo2 = 0;
    int split_passes = 0;            // search + reduce passes served (point targets)
    bool deep_tried = false;
    // point targets
    Geom<float> gf;
    PtF *pts = nullptr;        // cell-sorted (the NN search reads these 16-byte records)
    PtN *pn = nullptr;         // cell-sorted point + normal records (PlaneICP gathers these); NULL = no normals
    uint64_t serial = 0;
    // quirk Q6 (plane_icp.py:20-22: PlaneICP builds its tree on the ORIGINAL array, so a float64 target is searched in
    // float64 while the records are gathered from the float32 copy): the float64 coordinates of the same points, in the
    // index's cell-sorted order (w = original index), how far rounding moved any of them (metres), and the point grid's
    // geometry in double for the float64 box search.  nullptr: a float32 target (the PCD case)
    PtD *pts64 = nullptr;
    double band64 = 0;
    Geom<double> gq;
    // voxel targets
    Geom<double> gd;
    PtD *means = nullptr;      // cell-sorted
    double *vnorm = nullptr;   // [n][3] cell-sorted
    double *vicov = nullptr;   // [n][6] cell-sorted (xx xy xz yy yz zz)
    // voxel statistics in key order (device), for read-back
    double *st_mean = nullptr, *st_cov = nullptr, *st_norm = nullptr, *st_icov = nullptr;
    int64_t *st_counts = nullptr, *st_keys = nullptr;
    double voxel_size = 0;
    // float32 filter of the centroid search (k_nn_filter): a point index over the float32-ROUNDED centroids, in the
    // order of `means` (its "original index" is the cell-sorted index of the centroid); filter_band bounds how far a
    // centroid moves when it is rounded to float32 (metres).  nullptr: no filter (coordinates too large, PCR_VOX_FILTER=0)
    pcr_target *filter = nullptr;
    double filter_band = 0;
    int fused_passes = 0;          // fused small-scan passes served without a filter index (pass_setup)
    bool filter_tried = false;     // built by the first pass that can use it (search + reduce pipeline), not by set_target:
                                   // 0.3 ms that a 100 k-point scan -- fused kernel, float64 search -- never gets back
};

struct pcr_scan {
    pcr_context *ctx = nullptr;
    int64_t n = 0;
    float *x = nullptr, *y = nullptr, *z = nullptr;   // SoA, Morton-sorted unless PCR_FLAG_NO_SCAN_SORT
    // matched cell-sorted index per scan point (PCR_NONE = gated out), written by k_nn_scan and read
    // by the reduce kernel; kept across passes: the previous match is an exact upper bound for the
    // next search against the SAME target (nn_serial)
    uint32_t *nn_j = nullptr;
    uint64_t nn_serial = 0;
    // ---- certified reuse: state a pass leaves behind for the next one against the same target ----
    // after a TRACK / LIST pass: nn_j[i] = exact nearest neighbour (ungated; PCR_NONE = nothing inside the
    // search bound), lb2[i] = lower bound on the distance from the transformed point to every OTHER target point
    float *lb2 = nullptr;
    unsigned long long *umask = nullptr;   // 1 bit per scan point: not certified by k_certify -> search it
    uint32_t *ucnt = nullptr;              // per k_certify block: points marked
    int ucnt_cap = 0;
    bool pose_valid = false;               // prev_T = pose of the last pass over this scan
    double prev_T[16] = {0};
    bool track_valid = false;              // nn_j / lb2 hold tracking results for target `nn_serial` at prev_T
    float bb_c[3] = {0, 0, 0}, bb_e[3] = {0, 0, 0};   // bounding box of the scan: centre, half extents
    // statistics (pcr_scan_reuse_stats): passes by mode, points marked / examined by LIST passes, last pass
    int64_t st_passes[3] = {0, 0, 0};
    int64_t st_marked = 0, st_listed_of = 0;
    int last_mode = 0;
    int64_t last_marked = 0;               // (-1: a LIST pass whose count was not read back)
    double last_motion = -1;
    // every device block above, with its capacity: pcr_scan_destroy hands them back to the context's block cache
    // (seven hipFree calls used to be half of what `align(host array)` spends on upload + release, tools/align_seam_probe.py)
    std::vector<std::pair<void *, size_t>> blocks;
};
// device memory owned by a scan: from the context's block cache when a block fits, else hipMalloc
hipError_t pcr_scan_alloc(pcr_scan *s, void **p, size_t bytes);
void pcr_scan_free(pcr_scan *s, void *p);      // one block back to the cache (nullptr: no-op)

// what a pass does with the matches of the previous one
#define PCR_NN_FULL 0     // plain exact search of every point
#define PCR_NN_TRACK 1    // exact search of every point that also records the margin to the runner-up
#define PCR_NN_LIST 2     // k_certify proves most of the old matches still exact; tracking search of the rest

// ---- api.hip
void pcr_target_release(pcr_target *t);      // frees a target and everything it owns (blocks back to its context's cache)

// ---- index_build.hip
pcr_status pcr_build_point_grid(pcr_context *ctx, const float *d_xyz, int64_t n, float cell_hint, pcr_target *t, bool use_env = true,
                                double halo_default = 0.1);     // halo_default: margin of the extended lists, x cell (PCR_HALO overrides)
pcr_status pcr_build_centroid_filter(pcr_context *ctx, pcr_target *t);
pcr_status pcr_build_deep_lists(pcr_context *ctx, pcr_target *t);
pcr_status pcr_cell_population(pcr_context *ctx, pcr_target *t);      // fills pcr_target::pop_max / pop_p99 (once)
pcr_status pcr_build_centroid_grid(pcr_context *ctx, const double *d_mean, int64_t n, double cell, pcr_target *t);
pcr_status pcr_count_nonfinite(pcr_context *ctx, const void *d_xyz, int is_f64, int64_t n, int64_t *count, float *lo_out = nullptr,
                               float *hi_out = nullptr);     // (+ the bounding box of the finite points, rounded to float32)
pcr_status pcr_sort_scan(pcr_context *ctx, const float *d_xyz, int64_t n, unsigned flags, pcr_scan *s);
pcr_status pcr_permute_normals(pcr_context *ctx, const float *d_in, int64_t n, const PtF *pts, PtN *out);
pcr_status pcr_attach_points_f64(pcr_context *ctx, pcr_target *t, const double *d_xyz64);     // quirk Q6: pcr_target::pts64
pcr_status pcr_permute_rows_f64(pcr_context *ctx, const double *d_in, int64_t n, int in_stride, const int *cols,
                                int ncols, const PtD *means, double *out);

// ---- kernels.hip
pcr_status pcr_run_linearize(pcr_target *t, pcr_scan *s, int kind, const double T[16], double max_dist,
                             unsigned flags, double out[29]);
pcr_status pcr_run_align(pcr_target *t, pcr_scan *s, int kind, const double T_init[16], int max_iter, double tol,
                         double max_dist, unsigned flags, double T_out[16], int *iterations, double *trace_or_null);
pcr_status pcr_run_nn(pcr_target *t, const float *d_q, int64_t m, double r_max, void *d_dist, int64_t *d_idx, int f64);
pcr_status pcr_ensure_scratch(pcr_context *ctx, int64_t n_points);
bool pcr_pass_is_fused(const pcr_context *ctx, const pcr_scan *s);      // this scan runs the one-kernel (small-scan) form of a pass

// ---- roctx ranges around the hot-path launches (PCR_ROCTX=1; libroctx64 bound with dlopen, so the
// library loads without it).  Shows up in rocprofv3 --marker-trace / the tool's timeline.
void pcr_roctx_push(const char *name);
void pcr_roctx_pop();
struct RoctxRange {
    explicit RoctxRange(const char *name) { pcr_roctx_push(name); }
    ~RoctxRange() { pcr_roctx_pop(); }
};

// ---- profiling helpers (api.cpp)
void pcr_prof_begin(pcr_context *ctx, int kernel, ProfEvent *ev);
void pcr_prof_end(pcr_context *ctx, ProfEvent *ev);

// ---- comm.cpp
pcr_status pcr_comm_allreduce29(pcr_context *ctx, double *d_buf, PoseDev *pose = nullptr);
bool pcr_comm_failed_now(pcr_context *ctx);                       // a peer-to-peer exchange of this context timed out
pcr_status pcr_comm_p2p_local(pcr_context *const *members, int n);   // in-process peers (pcr_group)
