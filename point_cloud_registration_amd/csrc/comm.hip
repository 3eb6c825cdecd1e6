// Multi-GPU exchange step: one in-place ncclAllReduce(sum, float64, 29) per iteration on the
// context's stream.  The reference has no counterpart (single process).
//
// RCCL is bound at run time with dlopen so the library loads on hosts without RCCL and, in a
// process that already imported PyTorch, resolves to the very same librccl.so.1 (one RCCL per
// process).  232 bytes per message: latency-bound, xGMI bandwidth is irrelevant here.
#include <dlfcn.h>
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

extern "C" pcr_status pcr_comm_destroy(pcr_context *ctx) {
    if (!ctx || !ctx->comm) return PCR_OK;
    (void)hipStreamSynchronize(ctx->stream);
    if (g_nccl.CommDestroy) (void)g_nccl.CommDestroy((nccl_comm_t)ctx->comm);
    ctx->comm = nullptr; ctx->nranks = 1; ctx->rank = 0;
    return PCR_OK;
}

pcr_status pcr_comm_allreduce29(pcr_context *ctx, double *d_buf) {
    // ncclFloat64 = 8, ncclSum = 0 (rccl.h); in place, on the stream the kernels ran on
    NCCL_TRY(g_nccl.AllReduce(d_buf, d_buf, 29, 8, 0, (nccl_comm_t)ctx->comm, ctx->stream));
    return PCR_OK;
}
