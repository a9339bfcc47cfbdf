// amk_shard: the multi-GPU side of the batched-scenes sweep behind the C ABI.
//
// Scenes are independent MPC instances (the reference runs one: AM/src/mpc_obstacle_avoidance_node.cpp:8), so a sweep over
// many scenes is block-partitioned over the GPUs of a node -- one process per GPU, no data-path collective -- and the one
// exchange step is the gather of the per-scene controls (4 doubles) so that every rank holds all of them.  That gather is a
// direct ncclAllGather on RCCL (xGMI: every peer pair has its own link; the message is latency-bound).  RCCL is bound at
// run time (dlopen of librccl.so.1: the process may already carry PyTorch's copy, and a single-GPU host needs none).
#include "amk_common.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>

namespace {

struct Rccl {
    void *h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

Rccl &rccl() {
    static Rccl r;
    if (r.h) return r;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        r.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (r.h) break;
    }
    if (!r.h) return r;
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.h, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.h, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.h, "ncclCommDestroy");
    r.AllGather = (decltype(r.AllGather))dlsym(r.h, "ncclAllGather");
    r.AllReduce = (decltype(r.AllReduce))dlsym(r.h, "ncclAllReduce");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.h, "ncclGetErrorString");
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather && r.AllReduce;
    return r;
}

thread_local int g_last_nccl = 0;
int nccl_fail(ncclResult_t e) {
    g_last_nccl = (int)e;
    return AMK_ERR_HIP;
}
#define AMK_NCCL(expr)                                  \
    do {                                                \
        ncclResult_t _e = (expr);                       \
        if (_e != ncclSuccess) return nccl_fail(_e);    \
    } while (0)

}  // namespace

struct amk_shard {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
};

extern "C" {

int amk_shard_scene_range(int rank, int world, int total, int *first, int *count) {
    if (world <= 0 || rank < 0 || rank >= world || total < 0 || !first || !count) return AMK_ERR_INVALID_ARG;
    const int base = total / world, rem = total % world;   // contiguous blocks, sizes differ by at most one
    *first = rank * base + (rank < rem ? rank : rem);
    *count = base + (rank < rem ? 1 : 0);
    return AMK_OK;
}

int amk_shard_padded_count(int world, int total) {   // what every rank passes to the gathers: ncclAllGather needs equal counts
    if (world <= 0 || total < 0) return -1;
    return (total + world - 1) / world;
}

int amk_shard_unique_id(char *id_out) {
    static_assert(sizeof(ncclUniqueId) == AMK_SHARD_ID_BYTES, "ncclUniqueId size");
    if (!id_out) return AMK_ERR_INVALID_ARG;
    Rccl &r = rccl();
    if (!r.ok) return AMK_ERR_UNSUPPORTED;
    ncclUniqueId id;
    AMK_NCCL(r.GetUniqueId(&id));
    std::memcpy(id_out, &id, sizeof id);
    return AMK_OK;
}

int amk_shard_create(const char *id, int rank, int world, amk_shard **out) {
    if (!id || !out || world <= 0 || rank < 0 || rank >= world) return AMK_ERR_INVALID_ARG;
    *out = nullptr;
    if (amk_device_count() <= 0) return AMK_ERR_NO_DEVICE;
    Rccl &r = rccl();
    if (!r.ok) return AMK_ERR_UNSUPPORTED;
    ncclUniqueId nid;
    std::memcpy(&nid, id, sizeof nid);
    amk_shard *s = new amk_shard();
    s->rank = rank;
    s->world = world;
    const ncclResult_t e = r.CommInitRank(&s->comm, world, nid, rank);   // binds the calling thread's current HIP device
    if (e != ncclSuccess) {
        delete s;
        return nccl_fail(e);
    }
    *out = s;
    return AMK_OK;
}

int amk_shard_destroy(amk_shard *s) {
    if (!s) return AMK_ERR_INVALID_ARG;
    if (s->comm) (void)rccl().CommDestroy(s->comm);
    delete s;
    return AMK_OK;
}

int amk_shard_rank(const amk_shard *s) { return s ? s->rank : -1; }
int amk_shard_world(const amk_shard *s) { return s ? s->world : -1; }
int amk_shard_last_rccl_error(void) { return g_last_nccl; }

int amk_shard_gather(amk_shard *s, const double *d_local, long long n_doubles_per_rank, double *d_all, void *stream) {
    if (!s || !d_local || !d_all || n_doubles_per_rank < 0) return AMK_ERR_INVALID_ARG;
    if (n_doubles_per_rank == 0) return AMK_OK;
    AMK_NCCL(rccl().AllGather(d_local, d_all, (size_t)n_doubles_per_rank, ncclDouble, s->comm, (hipStream_t)stream));
    return AMK_OK;
}

int amk_shard_gather_u(amk_shard *s, const double *d_u_local, int n_local_scenes, double *d_u_all, void *stream) {
    return amk_shard_gather(s, d_u_local, 4LL * n_local_scenes, d_u_all, stream);
}

int amk_shard_max(amk_shard *s, double *d_values, int n, void *stream) {
    if (!s || !d_values || n <= 0) return AMK_ERR_INVALID_ARG;
    AMK_NCCL(rccl().AllReduce(d_values, d_values, (size_t)n, ncclDouble, ncclMax, s->comm, (hipStream_t)stream));
    return AMK_OK;
}

}  // extern "C"
