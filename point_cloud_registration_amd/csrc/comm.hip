// Multi-GPU exchange step: one in-place ncclAllReduce(sum, float64, 29) per iteration on the
// context's stream.  The reference has no counterpart (single process).
//
// RCCL is bound at run time with dlopen so the library loads on hosts without RCCL and, in a
// process that already imported PyTorch, resolves to the very same librccl.so.1 (one RCCL per
// process).  232 bytes per message: latency-bound, xGMI bandwidth is irrelevant here.
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

#include "pcr_internal.h"

typedef struct { char internal[128]; } nccl_uid;
typedef void *nccl_comm_t;
typedef int nccl_result;

static struct {
    void *lib;
    nccl_result (*GetUniqueId)(nccl_uid *);
    nccl_result (*CommInitRank)(nccl_comm_t *, int, nccl_uid, int);
    nccl_result (*CommDestroy)(nccl_comm_t);
    nccl_result (*AllReduce)(const void *, void *, size_t, int, int, nccl_comm_t, hipStream_t);
    const char *(*GetErrorString)(nccl_result);
} g_nccl;

static pcr_status load_rccl() {
    if (g_nccl.lib) return PCR_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *lib = nullptr;
    // RTLD_LOCAL on purpose: RCCL's dependency librocm_smi64.so must stay out of the global scope,
    // or a libamd_smi.so loaded later (e.g. by `import torch`) interposes its static maps onto it
    for (const char *n : names) { lib = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (lib) break; }
    if (!lib) { pcr_set_error("cannot load RCCL: %s", dlerror()); return PCR_ERR_COMM; }
    g_nccl.GetUniqueId = (decltype(g_nccl.GetUniqueId))dlsym(lib, "ncclGetUniqueId");
    g_nccl.CommInitRank = (decltype(g_nccl.CommInitRank))dlsym(lib, "ncclCommInitRank");
    g_nccl.CommDestroy = (decltype(g_nccl.CommDestroy))dlsym(lib, "ncclCommDestroy");
    g_nccl.AllReduce = (decltype(g_nccl.AllReduce))dlsym(lib, "ncclAllReduce");
    g_nccl.GetErrorString = (decltype(g_nccl.GetErrorString))dlsym(lib, "ncclGetErrorString");
    if (!g_nccl.GetUniqueId || !g_nccl.CommInitRank || !g_nccl.CommDestroy || !g_nccl.AllReduce) {
        pcr_set_error("RCCL library lacks a required symbol");
        return PCR_ERR_COMM;
    }
    g_nccl.lib = lib;
    return PCR_OK;
}

#define NCCL_TRY(expr)                                                                              \
    do {                                                                                            \
        nccl_result r_ = (expr);                                                                    \
        if (r_ != 0) {                                                                              \
            pcr_set_error("%s failed: %s", #expr, g_nccl.GetErrorString ? g_nccl.GetErrorString(r_) : "?"); \
            return PCR_ERR_COMM;                                                                    \
        }                                                                                           \
    } while (0)

extern "C" pcr_status pcr_comm_unique_id(void *id128) {
    PCR_REQUIRE(id128, "id128 is NULL");
    PCR_TRY(load_rccl());
    nccl_uid id;
    memset(&id, 0, sizeof id);
    NCCL_TRY(g_nccl.GetUniqueId(&id));
    memcpy(id128, &id, 128);
    return PCR_OK;
}

extern "C" pcr_status pcr_comm_init(pcr_context *ctx, const void *id128, int nranks, int rank) {
    PCR_REQUIRE(ctx && id128, "NULL argument");
    PCR_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "bad rank / nranks");
    PCR_REQUIRE(!ctx->comm, "communicator already initialised");
    PCR_TRY(load_rccl());
    HIP_TRY(hipSetDevice(ctx->device));
    nccl_uid id;
    memcpy(&id, id128, 128);
    nccl_comm_t comm = nullptr;
    NCCL_TRY(g_nccl.CommInitRank(&comm, nranks, id, rank));
    ctx->comm = comm; ctx->nranks = nranks; ctx->rank = rank;
    return PCR_OK;
}

// ---- peer-to-peer transport (round 5) -------------------------------------------------------------
// The same 232-byte exchange without a collective library in the loop: every rank owns a block of slots in device memory,
// exported with hipIpcGetMemHandle and mapped by its peers (one process per GPU).  After the fold a ONE-WAVE kernel stores
// the rank's 29 doubles into slot (seq mod 64, rank) of EVERY rank's block -- its own included -- drains the stores, then
// stores the sequence number behind them (system scope: sc0 sc1 payload -> vmcnt(0) -> release flag); it then spins on the
// nranks sequence words of ITS OWN block (acquire loads, a bounded number of polls) and sums the slots in rank order, so
// every rank computes bit-identical sums.  Every rank issues the same number of exchanges (the device-resident loop's
// top-up protocol guarantees it), so a slot is reused only 64 exchanges later, long after everybody read it.
// Status: exercised with two processes on ONE GPU (tests/test_gpu_two_ranks.py) -- which is also the first time the
// in-library N > 1 path of pcr_align (exchange between fold and k_gn_update, top-up of the queue) runs with N > 1 at all,
// RCCL refusing two ranks on one device.  Across xGMI the payload/flag ordering is the textbook one, but it has NOT been run
// on a multi-GPU node: opt-in (PCR_COMM=p2p), RCCL stays the default.
#define PCR_P2P_SLOTS 64
#define PCR_P2P_MAXR 8
#define PCR_P2P_BYTES ((size_t)PCR_P2P_SLOTS * PCR_P2P_MAXR * 32 * sizeof(double))
struct P2PState {
    double *own = nullptr;
    double *peer[PCR_P2P_MAXR] = {nullptr};
    bool opened[PCR_P2P_MAXR] = {false};
    int nranks = 1, rank = 0;
    unsigned long long seq = 0;
    int *h_err = nullptr;          // pinned + mapped: set by an exchange that gave up waiting (the host reads it without a copy)
    int *d_err = nullptr;          // its device-side address
    bool finegrained = false;
    bool local = false;            // peers live in THIS process (pcr_group): plain pointers, nothing to close
};
struct P2PArgs {
    double *peer[PCR_P2P_MAXR];
    int n, rank;
    unsigned long long seq;
    double *buf;
    int *err;
    PoseDev *pose;                 // device-resident loop: a failed exchange ends it on this rank (PCR_LOOP_COMMFAIL)
};

__global__ void __launch_bounds__(64) k_p2p_allreduce(const P2PArgs p) {
    const int l = threadIdx.x;
    const size_t slot = (size_t)(p.seq & (PCR_P2P_SLOTS - 1));
    const size_t mine = (slot * PCR_P2P_MAXR + (size_t)p.rank) * 32;
    if (l < 29) {
        const double v = p.buf[l];
        for (int r = 0; r < p.n; ++r) __hip_atomic_store(&p.peer[r][mine + l], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __threadfence_system();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the payload has left this wave before any flag does
    __builtin_amdgcn_wave_barrier();
    if (l < p.n)
        __hip_atomic_store(reinterpret_cast<unsigned long long *>(&p.peer[l][mine + 31]), p.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    double *own = p.peer[p.rank];
    bool ok = true;
    if (l < p.n) {
        const unsigned long long *flag = reinterpret_cast<const unsigned long long *>(&own[(slot * PCR_P2P_MAXR + (size_t)l) * 32 + 31]);
        // bounded by TIME (the constant-rate 100 MHz counter), not by a poll count: ~10 s, i.e. a peer died or never joined
        const unsigned long long t0 = wall_clock64();
        unsigned polls = 0;
        while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != p.seq) {
            if ((++polls & 1023u) == 0 && wall_clock64() - t0 > 1000000000ull) { ok = false; break; }
            __builtin_amdgcn_s_sleep(2);
        }
    }
    if (!__all(ok)) {
        // (a late or dead peer must not read as PCR_OK.  The flag is what pcr_linearize / pcr_align check; the
        // device-resident loop stops here, whatever the poisoned sums would have made k_gn_update do.)
        if (l == 0) {
            __hip_atomic_store(p.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (p.pose && p.pose->done == PCR_LOOP_RUNNING) p.pose->done = PCR_LOOP_COMMFAIL;
        }
        if (l < 29) p.buf[l] = __longlong_as_double(0x7ff8000000000000LL);      // the sums are NOT reduced: poison them
        return;
    }
    __threadfence_system();
    if (l < 29) {
        double s = 0.0;
        for (int r = 0; r < p.n; ++r)
            s += __hip_atomic_load(&own[(slot * PCR_P2P_MAXR + (size_t)r) * 32 + l], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        p.buf[l] = s;
    }
}

static void p2p_free(P2PState *st) {
    if (!st) return;
    if (!st->local) for (int r = 0; r < PCR_P2P_MAXR; ++r) if (st->opened[r]) (void)hipIpcCloseMemHandle(st->peer[r]);
    if (st->own) (void)hipFree(st->own);
    if (st->h_err) (void)hipHostFree(st->h_err);
    delete st;
}

// the slots of one rank + its error word.  fine-grained memory: peers' system-scope stores become visible to this GPU's
// loads without a kernel boundary; *handle (optional): the IPC handle of the block
static pcr_status p2p_state_create(pcr_context *ctx, hipIpcMemHandle_t *handle, P2PState **out) {
    P2PState *st = new P2PState();
    hipError_t e = hipExtMallocWithFlags((void **)&st->own, PCR_P2P_BYTES, hipDeviceMallocFinegrained);
    st->finegrained = e == hipSuccess;
    if (e != hipSuccess) { (void)hipGetLastError(); st->own = nullptr; e = hipMalloc((void **)&st->own, PCR_P2P_BYTES); }
    if (e != hipSuccess) { st->own = nullptr; p2p_free(st); pcr_set_error("p2p slots: %s", hipGetErrorString(e)); return PCR_ERR_HIP; }
    if (handle) {
        e = hipIpcGetMemHandle(handle, st->own);
        if (e != hipSuccess && st->finegrained) {                    // (some runtimes export coarse-grained blocks only)
            (void)hipGetLastError(); (void)hipFree(st->own); st->own = nullptr; st->finegrained = false;
            e = hipMalloc((void **)&st->own, PCR_P2P_BYTES);
            if (e != hipSuccess) st->own = nullptr;
            else e = hipIpcGetMemHandle(handle, st->own);
        }
        if (e != hipSuccess) {
            (void)hipGetLastError();
            p2p_free(st);
            pcr_set_error("hipIpcGetMemHandle: %s", hipGetErrorString(e));
            return PCR_ERR_COMM;
        }
    }
    // (coarse-grained memory is not coherent ACROSS devices while a kernel runs -- a spin on it may never see the
    // peer's flag.  Ranks that share one device go through its one L2 with system-scope accesses, which is the case the
    // fallback exists for (two processes on the test box); PCR_P2P_ALLOW_COARSE=0 refuses it, and pcr_comm_p2p_finegrained
    // lets the ranks agree before anybody exchanges: distributed.py takes the host transport when ranks on DIFFERENT devices
    // could not get fine-grained slots.)
    if (!st->finegrained) {
        const char *ac = getenv("PCR_P2P_ALLOW_COARSE");
        if (ac && atoi(ac) == 0) { p2p_free(st); pcr_set_error("p2p slots: no fine-grained device memory"); return PCR_ERR_COMM; }
    }
    hipError_t e2 = hipMemset(st->own, 0, PCR_P2P_BYTES);
    if (e2 == hipSuccess) e2 = hipHostMalloc((void **)&st->h_err, sizeof(int) * 16, hipHostMallocMapped | hipHostMallocCoherent);
    if (e2 == hipSuccess) { st->h_err[0] = 0; e2 = hipHostGetDevicePointer((void **)&st->d_err, st->h_err, 0); }
    if (e2 != hipSuccess) { (void)hipGetLastError(); p2p_free(st); pcr_set_error("p2p state: %s", hipGetErrorString(e2)); return PCR_ERR_HIP; }
    *out = st;
    return PCR_OK;
}

extern "C" pcr_status pcr_comm_p2p_export(pcr_context *ctx, void *handle64) {
    PCR_REQUIRE(ctx && handle64, "NULL argument");
    PCR_REQUIRE(!ctx->comm, "communicator already initialised");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "the boundary hands IPC handles around as 64 bytes");
    HIP_TRY(hipSetDevice(ctx->device));
    hipIpcMemHandle_t h;
    P2PState *st = nullptr;
    PCR_TRY(p2p_state_create(ctx, &h, &st));
    memcpy(handle64, &h, 64);
    ctx->comm = st; ctx->comm_kind = 1; ctx->nranks = 1; ctx->rank = 0;
    return PCR_OK;
}

// 1 when this context's slots are fine-grained (coherent across devices inside a kernel), 0 for the coarse-grained fallback
extern "C" pcr_status pcr_comm_p2p_finegrained(pcr_context *ctx, int *finegrained) {
    PCR_REQUIRE(ctx && finegrained, "NULL argument");
    *finegrained = (ctx->comm && ctx->comm_kind == 1 && ((P2PState *)ctx->comm)->finegrained) ? 1 : 0;
    return PCR_OK;
}

// ---- the same transport between contexts of ONE process (pcr_group, group.hip): peers are plain pointers ---------------
pcr_status pcr_comm_p2p_local(pcr_context *const *members, int n) {
    PCR_REQUIRE(members && n >= 1 && n <= PCR_P2P_MAXR, "a group has 1 to 8 members");
    if (n == 1) return PCR_OK;
    for (int i = 0; i < n; ++i) {
        pcr_context *ctx = members[i];
        PCR_REQUIRE(ctx && !ctx->comm, "member context already has a communicator");
        HIP_TRY(hipSetDevice(ctx->device));
        for (int j = 0; j < n; ++j) {
            if (members[j]->device == ctx->device) continue;
            const hipError_t e = hipDeviceEnablePeerAccess(members[j]->device, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) {
                pcr_set_error("hipDeviceEnablePeerAccess(%d -> %d): %s", ctx->device, members[j]->device, hipGetErrorString(e));
                (void)hipGetLastError();
                return PCR_ERR_COMM;
            }
            (void)hipGetLastError();
        }
        P2PState *st = nullptr;
        PCR_TRY(p2p_state_create(ctx, nullptr, &st));
        st->local = true;
        bool other_device = false;
        for (int j = 0; j < n; ++j) other_device |= members[j]->device != ctx->device;
        if (!st->finegrained && other_device) {
            p2p_free(st);
            pcr_set_error("peer-to-peer slots across devices need fine-grained device memory");
            return PCR_ERR_COMM;
        }
        ctx->comm = st; ctx->comm_kind = 1;
    }
    for (int i = 0; i < n; ++i) {
        P2PState *st = (P2PState *)members[i]->comm;
        for (int r = 0; r < n; ++r) st->peer[r] = ((P2PState *)members[r]->comm)->own;
        st->nranks = n; st->rank = i;
        members[i]->nranks = n; members[i]->rank = i;
    }
    return PCR_OK;
}

extern "C" pcr_status pcr_comm_p2p_attach(pcr_context *ctx, const void *handles, int nranks, int rank) {
    PCR_REQUIRE(ctx && handles, "NULL argument");
    PCR_REQUIRE(ctx->comm && ctx->comm_kind == 1, "pcr_comm_p2p_export first");
    PCR_REQUIRE(nranks >= 1 && nranks <= PCR_P2P_MAXR && rank >= 0 && rank < nranks, "bad rank / nranks (at most 8 ranks)");
    HIP_TRY(hipSetDevice(ctx->device));
    P2PState *st = (P2PState *)ctx->comm;
    for (int r = 0; r < nranks; ++r) {
        if (r == rank) { st->peer[r] = st->own; continue; }
        hipIpcMemHandle_t h;
        memcpy(&h, (const char *)handles + 64 * (size_t)r, 64);
        void *p = nullptr;
        const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) { (void)hipGetLastError(); pcr_set_error("hipIpcOpenMemHandle(rank %d): %s", r, hipGetErrorString(e)); return PCR_ERR_COMM; }
        st->peer[r] = (double *)p; st->opened[r] = true;
    }
    st->nranks = nranks; st->rank = rank;
    st->h_err[0] = 0;
    ctx->nranks = nranks; ctx->rank = rank;
    return PCR_OK;
}

extern "C" pcr_status pcr_comm_destroy(pcr_context *ctx) {
    if (!ctx || !ctx->comm) return PCR_OK;
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->comm_kind == 1) {
        p2p_free((P2PState *)ctx->comm);
    } else if (g_nccl.CommDestroy) {
        (void)g_nccl.CommDestroy((nccl_comm_t)ctx->comm);
    }
    ctx->comm = nullptr; ctx->comm_kind = 0; ctx->nranks = 1; ctx->rank = 0;
    return PCR_OK;
}

// 1 when a peer-to-peer exchange of this context gave up waiting for a peer (its sums were poisoned with NaN)
extern "C" pcr_status pcr_comm_p2p_failed(pcr_context *ctx, int *failed) {
    PCR_REQUIRE(ctx && failed, "NULL argument");
    *failed = 0;
    if (!ctx->comm || ctx->comm_kind != 1) return PCR_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    *failed = __atomic_load_n(((P2PState *)ctx->comm)->h_err, __ATOMIC_ACQUIRE) != 0 ? 1 : 0;
    return PCR_OK;
}

// the same word without a synchronisation: what pcr_linearize / pcr_align look at once the exchange they wait for is over
bool pcr_comm_failed_now(pcr_context *ctx) {
    if (!ctx->comm || ctx->comm_kind != 1) return false;
    return __atomic_load_n(((P2PState *)ctx->comm)->h_err, __ATOMIC_ACQUIRE) != 0;
}

pcr_status pcr_comm_allreduce29(pcr_context *ctx, double *d_buf, PoseDev *pose) {
    if (ctx->comm_kind == 1) {
        P2PState *st = (P2PState *)ctx->comm;
        P2PArgs a;
        for (int r = 0; r < PCR_P2P_MAXR; ++r) a.peer[r] = st->peer[r];
        a.n = st->nranks; a.rank = st->rank; a.seq = ++st->seq; a.buf = d_buf; a.err = st->d_err; a.pose = pose;
        hipLaunchKernelGGL(k_p2p_allreduce, dim3(1), dim3(64), 0, ctx->stream, a);
        HIP_TRY(hipGetLastError());
        return PCR_OK;
    }
    // ncclFloat64 = 8, ncclSum = 0 (rccl.h); in place, on the stream the kernels ran on
    NCCL_TRY(g_nccl.AllReduce(d_buf, d_buf, 29, 8, 0, (nccl_comm_t)ctx->comm, ctx->stream));
    return PCR_OK;
}
