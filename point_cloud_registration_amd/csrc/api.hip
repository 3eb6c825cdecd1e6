// C-ABI entry points of libpcr_hip.so (declared in include/pcr.h): contexts, targets, scans,
// the hot path, the behind-the-boundary Gauss-Newton driver, the KD-tree seam, profiling.
#include <math.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <string>

#include "gn_math.h"
#include "pcr_internal.h"

// ---- errors ---------------------------------------------------------------------------------
static thread_local std::string g_last_error;

void pcr_set_error(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

extern "C" const char *pcr_last_error(void) { return g_last_error.c_str(); }
extern "C" const char *pcr_version(void) { return "pcr-hip 0.4 (gfx950)"; }
extern "C" int pcr_abi_version(void) { return PCR_ABI_VERSION; }

extern "C" pcr_status pcr_device_count(int *count) {
    PCR_REQUIRE(count, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { n = 0; (void)hipGetLastError(); }
    *count = n;
    return PCR_OK;
}

// ---- cache of temporaries (pcr_internal.h) ----------------------------------------------------------
thread_local pcr_context *pcr_tls_ctx = nullptr;
typedef std::lock_guard<std::recursive_mutex> CacheLock;

void *pcr_cache_get(pcr_context *ctx, size_t bytes, size_t *cap_out, bool tight) {
    CacheLock lock(ctx->cache_mu);
    // smallest cached block that fits and is not wastefully large (<= 1.5x + 1 MiB; tight, for blocks that stay: 1.125x + 256 KiB)
    const size_t most = tight ? bytes + bytes / 8 + ((size_t)256 << 10) : bytes + bytes / 2 + ((size_t)1 << 20);
    int best = -1;
    for (int i = 0; i < (int)ctx->cache.size(); ++i) {
        const size_t c = ctx->cache[i].first;
        if (c >= bytes && c <= most && (best < 0 || c < ctx->cache[best].first)) best = i;
    }
    if (best < 0) return nullptr;
    void *p = ctx->cache[best].second;
    *cap_out = ctx->cache[best].first;
    ctx->cache_bytes -= ctx->cache[best].first;
    ctx->cache[best] = ctx->cache.back();
    ctx->cache.pop_back();
    return p;
}

void pcr_cache_put(pcr_context *ctx, void *p, size_t cap) {
    CacheLock lock(ctx->cache_mu);
    if (cap > ctx->cache_limit / 2 || ctx->cache.size() >= 256) { (void)hipFree(p); return; }
    while (ctx->cache_bytes + cap > ctx->cache_limit && !ctx->cache.empty()) {     // make room: drop the largest idle block
        int big = 0;
        for (int i = 1; i < (int)ctx->cache.size(); ++i) if (ctx->cache[i].first > ctx->cache[big].first) big = i;
        (void)hipFree(ctx->cache[big].second);
        ctx->cache_bytes -= ctx->cache[big].first;
        ctx->cache[big] = ctx->cache.back();
        ctx->cache.pop_back();
    }
    ctx->cache.push_back({cap, p});
    ctx->cache_bytes += cap;
}

void pcr_cache_clear(pcr_context *ctx) {
    CacheLock lock(ctx->cache_mu);
    for (auto &e : ctx->cache) (void)hipFree(e.second);
    ctx->cache.clear();
    ctx->cache_bytes = 0;
}

hipError_t pcr_persist_alloc(void **p, size_t bytes) {
    if (bytes == 0) bytes = 16;
    pcr_context *ctx = pcr_tls_ctx;
    if (!ctx) return hipMalloc(p, bytes);
    size_t cap = 0;
    *p = pcr_cache_get(ctx, bytes, &cap, true);
    if (!*p) {
        cap = (bytes + 4095) & ~(size_t)4095;
        const hipError_t e = pcr_malloc_retry(p, cap);
        if (e != hipSuccess) { *p = nullptr; return e; }
    }
    {
        CacheLock lock(ctx->cache_mu);
        ctx->owned[*p] = cap;
    }
    return hipSuccess;
}

void pcr_persist_free(pcr_context *ctx, void *p) {
    if (!p) return;
    if (ctx) {
        CacheLock lock(ctx->cache_mu);
        auto it = ctx->owned.find(p);
        if (it != ctx->owned.end()) {
            const size_t cap = it->second;
            ctx->owned.erase(it);
            pcr_cache_put(ctx, p, cap);      // reused only by this context's stream: ordered behind the old owner's kernels
            return;
        }
    }
    (void)hipFree(p);
}

hipError_t pcr_scan_alloc(pcr_scan *s, void **p, size_t bytes) {
    if (bytes == 0) bytes = 16;
    size_t cap = 0;
    *p = s->ctx ? pcr_cache_get(s->ctx, bytes, &cap) : nullptr;
    if (!*p) {
        cap = (bytes + 4095) & ~(size_t)4095;
        const hipError_t e = pcr_malloc_retry(p, cap);
        if (e != hipSuccess) { *p = nullptr; return e; }
    }
    s->blocks.push_back({*p, cap});
    return hipSuccess;
}

void pcr_scan_free(pcr_scan *s, void *p) {
    if (!p) return;
    for (size_t i = 0; i < s->blocks.size(); ++i) {
        if (s->blocks[i].first != p) continue;
        // the context's work is ordered on its one stream: a later user of the block queues behind this scan's kernels
        if (s->ctx) pcr_cache_put(s->ctx, p, s->blocks[i].second);
        else (void)hipFree(p);
        s->blocks[i] = s->blocks.back();
        s->blocks.pop_back();
        return;
    }
    (void)hipFree(p);
}

hipError_t pcr_malloc_retry(void **p, size_t bytes) {
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess && pcr_tls_ctx) {
        (void)hipGetLastError();
        pcr_cache_clear(pcr_tls_ctx);                 // (a no-op on an empty cache)
        e = hipMalloc(p, bytes);
    }
    return e;
}

// ---- context --------------------------------------------------------------------------------
extern "C" pcr_status pcr_context_create(int device, pcr_context **out) {
    PCR_REQUIRE(out, "out is NULL");
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) {
        pcr_set_error("device %d out of range (%d visible)", device, n);
        return PCR_ERR_INVALID;
    }
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    hipStream_t stream = nullptr;
    HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    pcr_context *ctx = new pcr_context();
    ctx->device = device;
    ctx->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    ctx->stream = stream;
    // 2 = per launch: search + reduce kernels for large scans (the fused kernel has half the occupancy),
    // ONE fused kernel for small, latency-bound ones (measured on MI355X)
    ctx->variant = 2;
    const char *v = getenv("PCR_VARIANT");
    if (v && *v) { const int vv = atoi(v); ctx->variant = vv == 0 ? 0 : (vv == 1 ? 1 : 2); }
    const char *nm = getenv("PCR_NN_MODE");
    if (nm && *nm) {
        // the values pcr_set_nn_mode accepts; anything else (1 was a search variant of round 2) -> the shipped search
        const int m = atoi(nm);
#ifdef PCR_DEV
        if (m == 0 || m == 2 || m == 3 || m == 4) ctx->nn_mode = m;
#else
        if (m == 0 || m == 3) ctx->nn_mode = m;
#endif
        else fprintf(stderr, "[pcr] PCR_NN_MODE=%s is not a search mode of this library (0, 3; developer build: 2, 4): using 0\n", nm);
    }
#ifdef PCR_DEV
    const char *ff = getenv("PCR_FUSE_FINALIZE");
    if (ff && *ff) ctx->fuse_finalize = atoi(ff) != 0;
#endif
    const char *lf = getenv("PCR_LOCAL_FRAC");
    if (lf && *lf) ctx->local_frac = atof(lf);
    const char *cm = getenv("PCR_VOXEL_CELL_MULT");
    if (cm && atof(cm) > 0) ctx->voxel_cell_mult = atof(cm);
    const char *vf = getenv("PCR_VOX_FILTER");
    if (vf && *vf) ctx->vox_filter = atoi(vf) != 0;
    const char *fa = getenv("PCR_FILTER_AFTER");
    if (fa && *fa) ctx->filter_after = atoi(fa);
    const char *vo = getenv("PCR_VOX_OCC");
    if (vo && *vo) ctx->vox_occ = atoi(vo) != 0;
    const char *psp = getenv("PCR_PHASE_SPLIT");
    if (psp && *psp) ctx->phase_split = atoi(psp) != 0;
    const char *tl = getenv("PCR_TILE_LOCAL");
    if (tl && *tl) ctx->tile_local = atoi(tl) != 0;
    const char *sd = getenv("PCR_STALL_DEBUG");
    if (sd && *sd) ctx->stall_debug = atoi(sd) != 0;
    const char *rp = getenv("PCR_RETIRE_PERIOD");
    if (rp && *rp) ctx->retire_period = atoi(rp);
    const char *ru = getenv("PCR_REUSE");
    if (ru && *ru) ctx->reuse = atoi(ru);
    const char *rt = getenv("PCR_REUSE_TAU");
    if (rt && *rt) ctx->reuse_tau = atof(rt);
    const char *rm = getenv("PCR_REUSE_MU");
    if (rm && *rm) ctx->reuse_mu = atof(rm);
    if (ctx->reuse < 0 || ctx->reuse > 2) ctx->reuse = 0;
    const char *cl = getenv("PCR_CACHE_LIMIT_MB");
    if (cl && *cl && atof(cl) >= 0) ctx->cache_limit = (size_t)(atof(cl) * 1048576.0);
    *out = ctx;
    return PCR_OK;
}

extern "C" pcr_status pcr_context_destroy(pcr_context *ctx) {
    if (!ctx) return PCR_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    pcr_comm_destroy(ctx);
    pcr_cache_clear(ctx);
    for (auto &e : ctx->prof_events) { (void)hipEventDestroy(e.start); (void)hipEventDestroy(e.stop); }
    for (auto &e : ctx->prof_free) { (void)hipEventDestroy(e.start); (void)hipEventDestroy(e.stop); }
    if (ctx->d_partials) (void)hipFree(ctx->d_partials);
    if (ctx->d_out) (void)hipFree(ctx->d_out);
    if (ctx->h_out) (void)hipHostFree(ctx->h_out);
    if (ctx->d_pose) (void)hipFree(ctx->d_pose);
    if (ctx->d_trace) (void)hipFree(ctx->d_trace);
    if (ctx->d_tile_ctr) (void)hipFree(ctx->d_tile_ctr);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return PCR_OK;
}

extern "C" pcr_status pcr_context_trim(pcr_context *ctx, uint64_t *released_bytes) {
    PCR_REQUIRE(ctx, "ctx is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));      // idle blocks may still be read by queued kernels of their last owner
    size_t before;
    { CacheLock lock(ctx->cache_mu); before = ctx->cache_bytes; }
    pcr_cache_clear(ctx);
    if (released_bytes) *released_bytes = (uint64_t)before;
    return PCR_OK;
}

extern "C" pcr_status pcr_context_stream(pcr_context *ctx, void **stream) {
    PCR_REQUIRE(ctx && stream, "NULL argument");
    *stream = (void *)ctx->stream;
    return PCR_OK;
}

extern "C" pcr_status pcr_context_synchronize(pcr_context *ctx) {
    PCR_REQUIRE(ctx, "ctx is NULL");
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return PCR_OK;
}

extern "C" pcr_status pcr_set_variant(pcr_context *ctx, int variant) {
    PCR_REQUIRE(ctx, "ctx is NULL");
    PCR_REQUIRE(variant >= 0 && variant <= 2, "variant must be 0, 1 or 2");
    ctx->variant = variant;
    return PCR_OK;
}

extern "C" pcr_status pcr_get_variant(pcr_context *ctx, int *variant) {
    PCR_REQUIRE(ctx && variant, "NULL argument");
    *variant = ctx->variant;
    return PCR_OK;
}

// ---- developer kernels (kernels_dev.hip): only in a library built with `make DEV=1` (libpcr_hip_dev.so) ----
extern "C" int pcr_has_dev_kernels(void) {
#ifdef PCR_DEV
    return 1;
#else
    return 0;
#endif
}
#ifndef PCR_DEV
extern "C" pcr_status pcr_nn_counters(pcr_target *, pcr_scan *, const double *, double, double *) {
    pcr_set_error("pcr_nn_counters needs the developer build of the library (make DEV=1: libpcr_hip_dev.so)");
    return PCR_ERR_INVALID;
}
#endif

extern "C" pcr_status pcr_set_fuse_finalize(pcr_context *ctx, int on) {
    PCR_REQUIRE(ctx, "ctx is NULL");
#ifndef PCR_DEV
    PCR_REQUIRE(on != 0, "the unfused fold kernels are in the developer build only (make DEV=1: libpcr_hip_dev.so)");
#endif
    ctx->fuse_finalize = on != 0;
    return PCR_OK;
}

extern "C" pcr_status pcr_get_pipeline(pcr_context *ctx, int *variant, int *fuse_finalize, int *nn_mode) {
    PCR_REQUIRE(ctx, "ctx is NULL");
    if (variant) *variant = ctx->variant;
    if (fuse_finalize) *fuse_finalize = ctx->fuse_finalize ? 1 : 0;
    if (nn_mode) *nn_mode = ctx->nn_mode;
    return PCR_OK;
}

extern "C" pcr_status pcr_set_nn_mode(pcr_context *ctx, int mode) {
    PCR_REQUIRE(ctx, "ctx is NULL");
    PCR_REQUIRE(mode == 0 || (mode >= 2 && mode <= 4), "nn mode must be 0, 2, 3 or 4");
#ifndef PCR_DEV
    PCR_REQUIRE(mode != 2 && mode != 4, "the wave-cooperative searches (2: LDS-staged, 4: MFMA-filtered) are in the developer build only (make dev: libpcr_hip_dev.so)");
#endif
    ctx->nn_mode = mode;
    return PCR_OK;
}

extern "C" pcr_status pcr_set_reuse(pcr_context *ctx, int mode, double tau, double mu) {
    PCR_REQUIRE(ctx, "ctx is NULL");
    PCR_REQUIRE(mode >= 0 && mode <= 2, "reuse mode must be 0 (off), 1 (automatic) or 2 (forced)");
    ctx->reuse = mode;
    if (tau > 0) ctx->reuse_tau = tau;
    if (mu > 0) ctx->reuse_mu = mu;
    return PCR_OK;
}

extern "C" pcr_status pcr_get_reuse(pcr_context *ctx, int *mode, double *tau, double *mu) {
    PCR_REQUIRE(ctx, "ctx is NULL");
    if (mode) *mode = ctx->reuse;
    if (tau) *tau = ctx->reuse_tau;
    if (mu) *mu = ctx->reuse_mu;
    return PCR_OK;
}

extern "C" pcr_status pcr_scan_reuse_stats(pcr_scan *s, double out[8]) {
    PCR_REQUIRE(s && out, "NULL argument");
    out[0] = (double)s->st_passes[0]; out[1] = (double)s->st_passes[1]; out[2] = (double)s->st_passes[2];
    out[3] = (double)s->st_marked; out[4] = (double)s->st_listed_of;
    out[5] = (double)s->last_mode; out[6] = (double)s->last_marked; out[7] = s->last_motion;
    return PCR_OK;
}

// ---- roctx ranges ------------------------------------------------------------------------------
#include <dlfcn.h>
static struct {
    int state;                 // 0 unknown, 1 active, -1 off
    int (*push)(const char *);
    int (*pop)();
} g_roctx;

static bool roctx_on() {
    if (g_roctx.state == 0) {
        g_roctx.state = -1;
        const char *e = getenv("PCR_ROCTX");
        if (e && atoi(e) != 0) {
            const char *names[] = {"libroctx64.so.4", "libroctx64.so", "/opt/rocm/lib/libroctx64.so"};
            void *lib = nullptr;
            for (const char *n : names) { lib = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (lib) break; }
            if (lib) {
                g_roctx.push = (int (*)(const char *))dlsym(lib, "roctxRangePushA");
                g_roctx.pop = (int (*)())dlsym(lib, "roctxRangePop");
                if (g_roctx.push && g_roctx.pop) g_roctx.state = 1;
            }
        }
    }
    return g_roctx.state == 1;
}
void pcr_roctx_push(const char *name) { if (roctx_on()) (void)g_roctx.push(name); }
void pcr_roctx_pop() { if (roctx_on()) (void)g_roctx.pop(); }

// ---- profiling: HIP events around every hot-path launch, on the launch stream ------------------
void pcr_prof_begin(pcr_context *ctx, int kernel, ProfEvent *ev) {
    ev->kernel = -1;
    if (!ctx->prof_on || !ctx->prof_this_pass) return;
    if (!ctx->prof_free.empty()) {
        *ev = ctx->prof_free.back();
        ctx->prof_free.pop_back();
    } else {
        if (hipEventCreate(&ev->start) != hipSuccess) return;
        if (hipEventCreate(&ev->stop) != hipSuccess) { (void)hipEventDestroy(ev->start); return; }
    }
    ev->kernel = kernel;
    (void)hipEventRecord(ev->start, ctx->stream);
}

void pcr_prof_end(pcr_context *ctx, ProfEvent *ev) {
    if (ev->kernel < 0) return;
    (void)hipEventRecord(ev->stop, ctx->stream);
    ctx->prof_events.push_back(*ev);
}

static void prof_drain(pcr_context *ctx) {
    (void)hipStreamSynchronize(ctx->stream);
    for (auto &e : ctx->prof_events) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e.start, e.stop) == hipSuccess) {
            ctx->prof_launches[e.kernel] += 1;
            ctx->prof_ms[e.kernel] += ms;
        }
        ctx->prof_free.push_back(e);
    }
    ctx->prof_events.clear();
}

extern "C" pcr_status pcr_profile_enable(pcr_context *ctx, int on) {
    PCR_REQUIRE(ctx, "ctx is NULL");
    if (!on) prof_drain(ctx);
    ctx->prof_on = on != 0;
    ctx->prof_period = on > 1 ? on : 1;        // on = n > 1: events around every n-th pass only (sampling)
    ctx->prof_pass = 0;
    ctx->prof_this_pass = ctx->prof_on;
    return PCR_OK;
}

extern "C" pcr_status pcr_profile_reset(pcr_context *ctx) {
    PCR_REQUIRE(ctx, "ctx is NULL");
    prof_drain(ctx);
    for (int i = 0; i < PCR_K_COUNT; ++i) { ctx->prof_launches[i] = 0; ctx->prof_ms[i] = 0; }
    return PCR_OK;
}

extern "C" pcr_status pcr_profile_read_n(pcr_context *ctx, int capacity, int64_t *launches, double *total_ms, int *count) {
    PCR_REQUIRE(ctx && launches && total_ms && capacity >= 0, "NULL argument");
    prof_drain(ctx);
    for (int i = 0; i < PCR_K_COUNT && i < capacity; ++i) { launches[i] = ctx->prof_launches[i]; total_ms[i] = ctx->prof_ms[i]; }
    if (count) *count = PCR_K_COUNT;
    return PCR_OK;
}

extern "C" pcr_status pcr_profile_read(pcr_context *ctx, int64_t launches[PCR_K_COUNT], double total_ms[PCR_K_COUNT]) {
    return pcr_profile_read_n(ctx, PCR_K_COUNT, launches, total_ms, nullptr);
}

// ---- small helpers ---------------------------------------------------------------------------
// sync = false: the caller's next step synchronises the stream before the entry point returns (the build that consumes the
// array does); a wait here only parks the device for the host's wake-up (35-45 us per set_target / scan set-up, round 5).  Such
// a caller passes its status through `synced`, which waits on the error paths that return before that step.
template <typename T>
static pcr_status upload(pcr_context *ctx, const T *host, size_t count, DevBuf<T> *buf, bool exact = false, bool sync = true) {
    HIP_TRY(exact ? buf->alloc_exact(count) : buf->alloc(count));
    if (count) {
        HIP_TRY(hipMemcpyAsync(buf->p, host, sizeof(T) * count, hipMemcpyHostToDevice, ctx->stream));
        if (sync) HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    return PCR_OK;
}
static pcr_status synced(pcr_context *ctx, pcr_status s) {
    if (s != PCR_OK) (void)hipStreamSynchronize(ctx->stream);      // (the host array may still be being read)
    return s;
}

void pcr_target_release(pcr_target *t);
static void target_free(pcr_target *t) { pcr_target_release(t); }
void pcr_target_release(pcr_target *t) {
    if (!t) return;
    target_free(t->filter);
    void *ptrs[] = {t->cell_start, t->cell_seed, t->rowocc, t->rbox, t->lbox, t->gbox, t->lbox_h, t->gbox_h, t->lbox_h2, t->gbox_h2, t->cs_h, t->pts_h, t->j_h, t->cs_h2, t->pts_h2, t->j_h2, t->pts, t->pn, t->pts64, t->means, t->vnorm, t->vicov,
                    t->st_mean, t->st_cov, t->st_norm, t->st_icov, t->st_counts, t->st_keys};
    if (t->ctx) (void)hipSetDevice(t->ctx->device);
    for (void *p : ptrs) pcr_persist_free(t->ctx, p);
    delete t;
}

// ---- point targets ---------------------------------------------------------------------------
static pcr_status points_create_common(pcr_context *ctx, const float *d_xyz, int64_t n, const float *d_normals,
                                       float cell_hint, pcr_target **out) {
    pcr_target *t = new pcr_target();
    t->ctx = ctx; t->is_voxel = 0; t->n = n; t->serial = ctx->next_serial++;
    pcr_status s = pcr_build_point_grid(ctx, d_xyz, n, cell_hint, t);
    if (s == PCR_OK && d_normals) {
        hipError_t e = pcr_persist_alloc((void **)&t->pn, sizeof(PtN) * (size_t)(n ? n : 1));
        if (e != hipSuccess) { pcr_set_error("hipMalloc normals: %s", hipGetErrorString(e)); s = PCR_ERR_HIP; }
        else s = pcr_permute_normals(ctx, d_normals, n, t->pts, t->pn);
        if (s == PCR_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) s = PCR_ERR_HIP;
    }
    if (s != PCR_OK) { target_free(t); return s; }
    *out = t;
    return PCR_OK;
}

extern "C" pcr_status pcr_target_points_create(pcr_context *ctx, const float *xyz, int64_t n,
                                               const float *normals_or_null, float cell_hint, pcr_target **out) {
    PCR_REQUIRE(ctx && out, "NULL argument");
    PCR_REQUIRE(n >= 0 && (xyz || n == 0), "bad point array");
    HIP_TRY(hipSetDevice(ctx->device));
    CtxScope scope(ctx);
    DevBuf<float> d_xyz, d_nrm;
    PCR_TRY(upload<float>(ctx, xyz, (size_t)n * 3, &d_xyz, false, false));
    if (normals_or_null) PCR_TRY(synced(ctx, upload<float>(ctx, normals_or_null, (size_t)n * 3, &d_nrm, false, false)));
    return synced(ctx, points_create_common(ctx, d_xyz.p, n, d_nrm.p, cell_hint, out));
}

extern "C" pcr_status pcr_target_points_create_device(pcr_context *ctx, const float *d_xyz, int64_t n,
                                                      const float *d_normals_or_null, float cell_hint,
                                                      pcr_target **out) {
    PCR_REQUIRE(ctx && out, "NULL argument");
    PCR_REQUIRE(n >= 0 && (d_xyz || n == 0), "bad point array");
    HIP_TRY(hipSetDevice(ctx->device));
    CtxScope scope(ctx);
    return points_create_common(ctx, d_xyz, n, d_normals_or_null, cell_hint, out);
}

extern "C" pcr_status pcr_target_set_normals(pcr_target *t, const float *normals) {
    PCR_REQUIRE(t && normals, "NULL argument");
    PCR_REQUIRE(!t->is_voxel, "normals belong to point targets");
    pcr_context *ctx = t->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    CtxScope scope(ctx);
    DevBuf<float> d_nrm;
    PCR_TRY(upload<float>(ctx, normals, (size_t)t->n * 3, &d_nrm));
    if (!t->pn) HIP_TRY(pcr_persist_alloc((void **)&t->pn, sizeof(PtN) * (size_t)(t->n ? t->n : 1)));
    PCR_TRY(pcr_permute_normals(ctx, d_nrm.p, t->n, t->pts, t->pn));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return PCR_OK;
}

// quirk Q6 (plane_icp.py:20-22): the float64 coordinates of the points the target was created from
extern "C" pcr_status pcr_target_points_set_f64(pcr_target *t, const double *xyz64) {
    PCR_REQUIRE(t && xyz64, "NULL argument");
    PCR_REQUIRE(!t->is_voxel, "float64 search coordinates belong to point targets");
    pcr_context *ctx = t->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    CtxScope scope(ctx);
    DevBuf<double> d_xyz;
    PCR_TRY(upload<double>(ctx, xyz64, (size_t)t->n * 3, &d_xyz));
    return pcr_attach_points_f64(ctx, t, d_xyz.p);
}

__global__ void __launch_bounds__(256) k_unpermute_normals(const PtN *__restrict__ pn, int64_t n, float *out) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const PtN v = pn[j];
    const size_t i = v.orig;
    out[3 * i] = v.nx; out[3 * i + 1] = v.ny; out[3 * i + 2] = v.nz;
}

extern "C" pcr_status pcr_target_get_normals(pcr_target *t, float *normals_out) {
    PCR_REQUIRE(t && normals_out, "NULL argument");
    if (t->is_voxel || !t->pn) { pcr_set_error("target has no per-point normals"); return PCR_ERR_NO_TARGET; }
    pcr_context *ctx = t->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    if (t->n == 0) return PCR_OK;
    CtxScope scope(ctx);
    DevBuf<float> d_out;
    HIP_TRY(d_out.alloc(3 * (size_t)t->n));
    hipLaunchKernelGGL(k_unpermute_normals, dim3((unsigned)((t->n + 255) / 256)), dim3(256), 0, ctx->stream,
                       t->pn, t->n, d_out.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(normals_out, d_out.p, sizeof(float) * 3 * (size_t)t->n, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return PCR_OK;
}

// ---- voxel targets from statistics -----------------------------------------------------------
pcr_status pcr_voxel_target_finish(pcr_context *ctx, pcr_target *t, double voxel_size) {
    // t->st_mean (+ st_norm, st_icov) are on the device in key order: build the centroid grid and
    // the cell-sorted copies the kernels gather from.
    t->voxel_size = voxel_size;
    // Cell edge = 2 voxels: about four centroids per occupied cell on a surface, the occupancy
    // the point grid's automatic cell size aims for too.  Measured (vplane_10m, search us per
    // pose): 1x 940, 1.5x 880, 2x 830, 2.5x 1020, 3x 1150; ndt_10m totals are equal at 1x and 2x.
    PCR_TRY(pcr_build_centroid_grid(ctx, t->st_mean, t->n, voxel_size * ctx->voxel_cell_mult, t));
    const size_t nn = (size_t)(t->n ? t->n : 1);
    if (t->st_norm) {
        HIP_TRY(pcr_persist_alloc((void **)&t->vnorm, sizeof(double) * 3 * nn));
        const int cols[3] = {0, 1, 2};
        PCR_TRY(pcr_permute_rows_f64(ctx, t->st_norm, t->n, 3, cols, 3, t->means, t->vnorm));
    }
    if (t->st_icov) {
        HIP_TRY(pcr_persist_alloc((void **)&t->vicov, sizeof(double) * 6 * nn));
        const int cols[6] = {0, 1, 2, 4, 5, 8};
        PCR_TRY(pcr_permute_rows_f64(ctx, t->st_icov, t->n, 9, cols, 6, t->means, t->vicov));
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return PCR_OK;
}

extern "C" pcr_status pcr_target_voxels_create_from_stats(pcr_context *ctx, const double *mean,
                                                          const double *norm_or_null, const double *icov_or_null,
                                                          int64_t n_v, double voxel_size, pcr_target **out) {
    PCR_REQUIRE(ctx && out, "NULL argument");
    PCR_REQUIRE(n_v >= 0 && (mean || n_v == 0), "bad mean array");
    PCR_REQUIRE(voxel_size > 0, "voxel_size must be positive");
    HIP_TRY(hipSetDevice(ctx->device));
    pcr_target *t = new pcr_target();
    t->ctx = ctx; t->is_voxel = 1; t->n = n_v; t->serial = ctx->next_serial++;
    CtxScope scope(ctx);
    DevBuf<double> b_mean, b_norm, b_icov;
    pcr_status s = upload<double>(ctx, mean, (size_t)n_v * 3, &b_mean, true);
    if (s == PCR_OK && norm_or_null) s = upload<double>(ctx, norm_or_null, (size_t)n_v * 3, &b_norm, true);
    if (s == PCR_OK && icov_or_null) s = upload<double>(ctx, icov_or_null, (size_t)n_v * 9, &b_icov, true);
    t->st_mean = b_mean.release(); t->st_norm = b_norm.release(); t->st_icov = b_icov.release();
    if (s == PCR_OK) s = pcr_voxel_target_finish(ctx, t, voxel_size);
    if (s != PCR_OK) { target_free(t); return s; }
    *out = t;
    return PCR_OK;
}

extern "C" pcr_status pcr_target_voxels_get(pcr_target *t, int64_t *n_v, double *mean, double *cov, double *norm,
                                            double *icov, int64_t *counts, int64_t *keys) {
    PCR_REQUIRE(t && n_v, "NULL argument");
    PCR_REQUIRE(t->is_voxel, "not a voxel target");
    pcr_context *ctx = t->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    *n_v = t->n;
    const size_t n = (size_t)t->n;
    struct { void *dst; const void *src; size_t bytes; const char *what; } c[] = {
        {mean, t->st_mean, n * 3 * 8, "mean"}, {cov, t->st_cov, n * 9 * 8, "cov"}, {norm, t->st_norm, n * 3 * 8, "norm"},
        {icov, t->st_icov, n * 9 * 8, "icov"}, {counts, t->st_counts, n * 8, "counts"}, {keys, t->st_keys, n * 8, "keys"}};
    for (auto &e : c) {
        if (!e.dst) continue;
        if (!e.src) { pcr_set_error("voxel target does not hold '%s'", e.what); return PCR_ERR_NO_TARGET; }
        if (e.bytes) HIP_TRY(hipMemcpyAsync(e.dst, e.src, e.bytes, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return PCR_OK;
}

extern "C" pcr_status pcr_target_size(pcr_target *t, int64_t *n) {
    PCR_REQUIRE(t && n, "NULL argument");
    *n = t->n;
    return PCR_OK;
}

extern "C" pcr_status pcr_target_index_info(pcr_target *t, double *cell, int64_t dims[3], int64_t *occupied, int64_t *n) {
    PCR_REQUIRE(t, "NULL argument");
    if (cell) *cell = t->is_voxel ? t->gd.h : (double)t->gf.h;
    if (dims) {
        dims[0] = t->is_voxel ? t->gd.nx : t->gf.nx;
        dims[1] = t->is_voxel ? t->gd.ny : t->gf.ny;
        dims[2] = t->is_voxel ? t->gd.nz : t->gf.nz;
    }
    if (occupied) *occupied = t->occupied;
    if (n) *n = t->n;
    return PCR_OK;
}

extern "C" pcr_status pcr_target_index_population(pcr_target *t, int64_t *pop_max, int64_t *pop_p99, int *heavy) {
    PCR_REQUIRE(t, "NULL argument");
    if (pop_max || pop_p99) {
        HIP_TRY(hipSetDevice(t->ctx->device));
        CtxScope scope(t->ctx);
        PCR_TRY(pcr_cell_population(t->ctx, t));
    }
    if (pop_max) *pop_max = t->pop_max;
    if (pop_p99) *pop_p99 = t->pop_p99;
    if (heavy) *heavy = t->heavy ? 1 : 0;
    return PCR_OK;
}

extern "C" pcr_status pcr_target_index_halo(pcr_target *t, double *halo, int64_t *records) {
    PCR_REQUIRE(t, "NULL argument");
    const pcr_target *p = t->is_voxel ? t->filter : t;      // voxel targets: the float32 filter index of the centroid search
    if (halo) *halo = p ? (double)p->gf.halo : 0.0;
    if (records) *records = p ? p->n_h : 0;
    return PCR_OK;
}

extern "C" pcr_status pcr_target_index_halo2(pcr_target *t, double *halo, int64_t *records) {
    PCR_REQUIRE(t, "NULL argument");
    if (halo) *halo = t->is_voxel ? 0.0 : (double)t->halo2;
    if (records) *records = t->is_voxel ? 0 : t->n_h2;
    return PCR_OK;
}

extern "C" pcr_status pcr_target_filter_band(pcr_target *t, double *band) {
    PCR_REQUIRE(t && band, "NULL argument");
    *band = (t->is_voxel && t->filter) ? t->filter_band : 0.0;
    return PCR_OK;
}

extern "C" pcr_status pcr_target_destroy(pcr_target *t) {
    if (!t) return PCR_OK;
    (void)hipSetDevice(t->ctx->device);
    (void)hipStreamSynchronize(t->ctx->stream);
    target_free(t);
    return PCR_OK;
}

// ---- scan ------------------------------------------------------------------------------------
extern "C" pcr_status pcr_scan_create_device(pcr_context *ctx, const float *d_xyz, int64_t n, unsigned flags,
                                             pcr_scan **out) {
    PCR_REQUIRE(ctx && out, "NULL argument");
    PCR_REQUIRE(n >= 0 && (d_xyz || n == 0), "bad scan array");
    CtxScope scope(ctx);
    pcr_scan *s = new pcr_scan();
    s->ctx = ctx;
    pcr_status st = pcr_sort_scan(ctx, d_xyz, n, flags, s);
    if (st != PCR_OK) { pcr_scan_destroy(s); return st; }
    *out = s;
    return PCR_OK;
}

extern "C" pcr_status pcr_scan_create(pcr_context *ctx, const float *xyz, int64_t n, unsigned flags, pcr_scan **out) {
    PCR_REQUIRE(ctx && out, "NULL argument");
    PCR_REQUIRE(n >= 0 && (xyz || n == 0), "bad scan array");
    HIP_TRY(hipSetDevice(ctx->device));
    CtxScope scope(ctx);
    DevBuf<float> d_xyz;
    PCR_TRY(upload<float>(ctx, xyz, (size_t)n * 3, &d_xyz, false, false));
    return synced(ctx, pcr_scan_create_device(ctx, d_xyz.p, n, flags, out));
}

extern "C" pcr_status pcr_scan_size(pcr_scan *s, int64_t *n) {
    PCR_REQUIRE(s && n, "NULL argument");
    *n = s->n;
    return PCR_OK;
}

// test / diagnostic seam: the correspondences the last search + reduce pass left behind, in the scan's device order
extern "C" pcr_status pcr_scan_read_matches(pcr_scan *s, uint32_t *out) {
    PCR_REQUIRE(s && (out || s->n == 0), "NULL argument");
    if (!s->nn_j || s->nn_serial == 0) { pcr_set_error("no search + reduce pass has run over this scan"); return PCR_ERR_INVALID; }
    HIP_TRY(hipSetDevice(s->ctx->device));
    if (s->n == 0) return PCR_OK;
    HIP_TRY(hipMemcpyAsync(out, s->nn_j, sizeof(uint32_t) * (size_t)s->n, hipMemcpyDeviceToHost, s->ctx->stream));
    HIP_TRY(hipStreamSynchronize(s->ctx->stream));
    return PCR_OK;
}

extern "C" pcr_status pcr_scan_destroy(pcr_scan *s) {
    if (!s) return PCR_OK;
    if (s->ctx) (void)hipSetDevice(s->ctx->device);
    // no stream synchronisation and no hipFree: the blocks go back to the context's cache, whose next user runs on
    // the same stream behind whatever of this scan is still queued (e.g. the no-op tail of a device-resident loop)
    while (!s->blocks.empty()) pcr_scan_free(s, s->blocks.back().first);
    delete s;
    return PCR_OK;
}

// ---- hot path --------------------------------------------------------------------------------
extern "C" pcr_status pcr_linearize(pcr_target *t, pcr_scan *s, int kind, const double T[16], double max_dist,
                                    unsigned flags, double out[29]) {
    PCR_REQUIRE(t && s && T && out, "NULL argument");
    CtxScope scope(t->ctx);
    return pcr_run_linearize(t, s, kind, T, max_dist, flags, out);
}

// ---- Gauss-Newton driver behind the boundary (registration.py:71-113) -------------------------
// Default: the device-resident loop (kernels.hip: pcr_run_align).  PCR_FLAG_HOST_LOOP keeps the
// host-driven form (one pcr_linearize + host solve per iteration), the same arithmetic from gn_math.h.
static pcr_status align_host_loop(pcr_target *t, pcr_scan *s, int kind, const double T_init[16], int max_iter, double tol,
                                  double max_dist, unsigned flags, double T_out[16], int *iterations,
                                  double *trace_or_null) {
    double T[16];
    memcpy(T, T_init, sizeof T);
    int it = 0;
    for (; it < max_iter; ++it) {
        double o[29];
        PCR_TRY(pcr_run_linearize(t, s, kind, T, max_dist, flags, o));
        if (trace_or_null) {
            memcpy(trace_or_null + (size_t)it * 45, T, 16 * sizeof(double));
            memcpy(trace_or_null + (size_t)it * 45 + 16, o, 29 * sizeof(double));
        }
        double A[6][7];
        const int r = gn_step(A, o, tol, T);
        if (r == 2) {
            pcr_set_error("Singular matrix (correspondences: %.0f)", o[28]);
            if (iterations) *iterations = it + 1;
            memcpy(T_out, T, sizeof T);
            return PCR_ERR_SINGULAR;
        }
        if (r == 1) { ++it; break; }              // registration.py:106-108: test precedes the update (Q4)
    }
    if (iterations) *iterations = it;
    memcpy(T_out, T, sizeof T);
    return PCR_OK;
}

extern "C" pcr_status pcr_align(pcr_target *t, pcr_scan *s, int kind, const double T_init[16], int max_iter, double tol,
                                double max_dist, unsigned flags, double T_out[16], int *iterations,
                                double *trace_or_null) {
    PCR_REQUIRE(t && s && T_init && T_out, "NULL argument");
    CtxScope scope(t->ctx);
    // (certified reuse runs on host-driven passes; forced on, the loop is driven from the host.  Small scans on one GPU:
    // the zero-copy hand-off makes the host-driven iteration cheaper than the device-resident one -- one kernel either
    // way, but the step costs ~3 us on one GPU thread against < 1 us on the host)
    const bool use_comm = t->ctx->comm != nullptr && !(flags & PCR_FLAG_LOCAL_ONLY);
    const bool small_host = !(flags & PCR_FLAG_DEVICE_LOOP) && !use_comm && pcr_pass_is_fused(t->ctx, s);
    if ((flags & PCR_FLAG_HOST_LOOP) || t->ctx->reuse == 2 || small_host)
        return align_host_loop(t, s, kind, T_init, max_iter, tol, max_dist, flags, T_out, iterations, trace_or_null);
    return pcr_run_align(t, s, kind, T_init, max_iter, tol, max_dist, flags, T_out, iterations, trace_or_null);
}

// ---- KD-tree seam ----------------------------------------------------------------------------
static pcr_status nn_query_common(pcr_target *t, const float *q, int64_t m, double r_max, void *dist, int64_t *idx, int f64) {
    PCR_REQUIRE(t && (q || m == 0) && (dist || m == 0) && (idx || m == 0), "NULL argument");
    pcr_context *ctx = t->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    if (m == 0) return PCR_OK;
    CtxScope scope(ctx);
    DevBuf<float> d_q;
    PCR_TRY(upload<float>(ctx, q, (size_t)m * 3, &d_q));
    const size_t ds = f64 ? 8 : 4;
    DevBuf<char> d_dist;
    DevBuf<int64_t> d_idx;
    if (d_dist.alloc_bytes(ds * (size_t)m) != hipSuccess || d_idx.alloc((size_t)m) != hipSuccess) {
        pcr_set_error("hipMalloc failed for %lld query results", (long long)m);
        return PCR_ERR_NOMEM;
    }
    PCR_TRY(pcr_run_nn(t, d_q.p, m, r_max, d_dist.p, d_idx.p, f64));
    if (hipMemcpyAsync(dist, d_dist.p, ds * (size_t)m, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
        hipMemcpyAsync(idx, d_idx.p, 8 * (size_t)m, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess) {
        pcr_set_error("copy-back of query results failed: %s", hipGetErrorString(hipGetLastError()));
        return PCR_ERR_HIP;
    }
    return PCR_OK;
}

extern "C" pcr_status pcr_nn_query(pcr_target *t, const float *q, int64_t m, float r_max, float *dist, int64_t *idx) {
    return nn_query_common(t, q, m, (double)r_max, dist, idx, 0);
}

extern "C" pcr_status pcr_nn_query_f64(pcr_target *t, const float *q, int64_t m, double r_max, double *dist, int64_t *idx) {
    return nn_query_common(t, q, m, r_max, dist, idx, 1);
}
