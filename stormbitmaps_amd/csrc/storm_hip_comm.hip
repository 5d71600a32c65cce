// storm_hip_comm.hip — the one inter-GPU exchange of the all-pairs path for C callers: an 8-byte sum over
// the ranks of a multi-process run (one process per GPU), through RCCL over xGMI.
//
// The reference has no such step (one process, one thread: SURVEY §5 "distributed communication"); the device
// build shards the pair space over ranks (STORM_hip_set_shard / storm_hip_pairw_dense(..., rank, world)) and
// the partial totals have to be added: ncclAllReduce(count = 1, ncclUint64, ncclSum). bench.py does it
// through torch.distributed (backend "nccl" = RCCL); a C host has this shim. librccl is loaded on first use
// (dlopen), so single-GPU users need nothing installed beyond the HIP runtime.
#include <dlfcn.h>

#include <cstring>

#include "storm_hip_internal.h"

using namespace storm;

struct storm_hip_comm_s {
    void* comm = nullptr;   // ncclComm_t
    uint32_t rank = 0, world = 1;
    unsigned long long* d_word = nullptr;
    unsigned long long* h_word = nullptr;  // pinned
};

namespace {

struct Rccl {
    struct Id { char b[128]; };  // ncclUniqueId, passed by value
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, Id, int) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool tried = false;
};
Rccl g_rccl;
constexpr int kNcclUint64 = 5, kNcclSum = 0;  // rccl.h: ncclDataType_t / ncclRedOp_t

bool load_rccl() {
    if (g_rccl.lib) return true;
    if (g_rccl.tried) return false;
    g_rccl.tried = true;
    // STORM_HIP_RCCL names another build (e.g. the one a PyTorch wheel ships); RTLD_GLOBAL is not needed
    const char* names[] = {getenv("STORM_HIP_RCCL"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        if (!n || !n[0]) continue;
        if ((g_rccl.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    }
    if (!g_rccl.lib) {
        set_error("RCCL: librccl.so not found (%s)", dlerror());
        return false;
    }
    auto sym = [&](const char* name) { return dlsym(g_rccl.lib, name); };
    g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(sym("ncclGetUniqueId"));
    g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(sym("ncclCommInitRank"));
    g_rccl.AllReduce = reinterpret_cast<decltype(g_rccl.AllReduce)>(sym("ncclAllReduce"));
    g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(sym("ncclCommDestroy"));
    g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(sym("ncclGetErrorString"));
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.CommDestroy) {
        set_error("RCCL: librccl.so lacks an entry point this shim needs");
        dlclose(g_rccl.lib);
        g_rccl.lib = nullptr;
        return false;
    }
    return true;
}

int rccl_fail(const char* what, int rc) {
    set_error("RCCL: %s failed: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?");
    return STORM_HIP_EHIP;
}

}  // namespace

extern "C" {

int storm_hip_comm_unique_id(uint8_t id[STORM_HIP_COMM_ID_BYTES]) {
    if (!id) {
        set_error("comm_unique_id: NULL buffer");
        return STORM_HIP_EINVAL;
    }
    if (!load_rccl()) return STORM_HIP_ENODEV;
    if (int rc = g_rccl.GetUniqueId(id)) return rccl_fail("ncclGetUniqueId", rc);
    return STORM_HIP_OK;
}

int storm_hip_comm_init_rank(storm_hip_ctx_t* ctx, const uint8_t id[STORM_HIP_COMM_ID_BYTES], uint32_t rank,
                             uint32_t world, storm_hip_comm_t** out) {
    if (!ctx || !id || !out || world == 0 || rank >= world) {
        set_error("comm_init_rank: bad arguments");
        return STORM_HIP_EINVAL;
    }
    *out = nullptr;
    if (!load_rccl()) return STORM_HIP_ENODEV;
    STORM_HIP_TRY(hipSetDevice(ctx->device));
    storm_hip_comm_t* c = new (std::nothrow) storm_hip_comm_t();
    if (!c) return STORM_HIP_ENOMEM;
    c->rank = rank;
    c->world = world;
    Rccl::Id uid;
    memcpy(uid.b, id, sizeof(uid.b));
    int rc = g_rccl.CommInitRank(&c->comm, (int)world, uid, (int)rank);
    if (rc) {
        delete c;
        return rccl_fail("ncclCommInitRank", rc);
    }
    if (hipMalloc(reinterpret_cast<void**>(&c->d_word), 64) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void**>(&c->h_word), 64, hipHostMallocDefault) != hipSuccess) {
        storm_hip_comm_destroy(c);
        set_error("comm_init_rank: allocation of the exchange word failed");
        return STORM_HIP_ENOMEM;
    }
    *out = c;
    return STORM_HIP_OK;
}

// In place: *value becomes the sum of every rank's *value. Ordered on the context's stream behind
// whatever produced the value there; returns when the sum is on the host.
int storm_hip_comm_allreduce_u64(storm_hip_ctx_t* ctx, storm_hip_comm_t* comm, uint64_t* value) {
    return storm_hip_comm_allreduce_u64s(ctx, comm, value, 1);
}

// The same for up to 8 words at once (one collective): e.g. {partial total, failure flag}, so that a rank whose
// pass failed still enters the collective and every rank learns of the failure.
int storm_hip_comm_allreduce_u64s(storm_hip_ctx_t* ctx, storm_hip_comm_t* comm, uint64_t* values, uint32_t n) {
    if (!ctx || !comm || !values || n == 0 || n > 8) {
        set_error("comm_allreduce_u64s: NULL argument or a word count outside 1..8");
        return STORM_HIP_EINVAL;
    }
    STORM_HIP_TRY(hipSetDevice(ctx->device));
    memcpy(comm->h_word, values, n * sizeof(uint64_t));
    STORM_HIP_TRY(hipMemcpyAsync(comm->d_word, comm->h_word, n * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
    if (int rc = g_rccl.AllReduce(comm->d_word, comm->d_word, n, kNcclUint64, kNcclSum, comm->comm, ctx->stream))
        return rccl_fail("ncclAllReduce", rc);
    STORM_HIP_TRY(hipMemcpyAsync(comm->h_word, comm->d_word, n * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    STORM_HIP_TRY(hipStreamSynchronize(ctx->stream));
    memcpy(values, comm->h_word, n * sizeof(uint64_t));
    return STORM_HIP_OK;
}

// The shard's partial straight from the context's result word (storm_hip_pairw_dense_begin /
// storm_hip_pairw_sparse_begin leave it there): all-reduced on the device, one host wait for pass + sum.
int storm_hip_comm_allreduce_result(storm_hip_ctx_t* ctx, storm_hip_comm_t* comm, uint64_t* total) {
    if (!ctx || !comm || !total) {
        set_error("comm_allreduce_result: NULL argument");
        return STORM_HIP_EINVAL;
    }
    STORM_HIP_TRY(hipSetDevice(ctx->device));
    if (ctx->mail_armed) {   // the pass was launched into the host mailbox (option result_mailbox): its value goes up again
        uint64_t mine = 0;
        if (int rc = storm::wait_mailbox(ctx, &mine)) return rc;
        return storm_hip_comm_allreduce_u64(ctx, comm, &mine) ? STORM_HIP_EHIP : ((*total = mine), STORM_HIP_OK);
    }
    if (int rc = g_rccl.AllReduce(ctx->d_scalar, comm->d_word, 1, kNcclUint64, kNcclSum, comm->comm, ctx->stream))
        return rccl_fail("ncclAllReduce", rc);
    STORM_HIP_TRY(hipMemcpyAsync(comm->h_word, comm->d_word, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    STORM_HIP_TRY(hipStreamSynchronize(ctx->stream));
    *total = *comm->h_word;
    return STORM_HIP_OK;
}

uint32_t storm_hip_comm_rank(const storm_hip_comm_t* comm) { return comm ? comm->rank : 0; }
uint32_t storm_hip_comm_world(const storm_hip_comm_t* comm) { return comm ? comm->world : 0; }

void storm_hip_comm_destroy(storm_hip_comm_t* comm) {
    if (!comm) return;
    if (comm->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(comm->comm);
    if (comm->d_word) (void)hipFree(comm->d_word);
    if (comm->h_word) (void)hipHostFree(comm->h_word);
    delete comm;
}

}  // extern "C"
