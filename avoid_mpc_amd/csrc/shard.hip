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

#include <chrono>
#include <cstring>
#include <string>
#include <thread>

namespace {

struct Rccl {
    void *h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;
    bool ok = false;
    bool was_loaded = false;   // the library was in the process already (RTLD_NOLOAD found it): e.g. PyTorch's bundled copy
    std::string path;          // where the bound ncclAllGather lives (dladdr)
};

Rccl &rccl() {
    static Rccl r;
    if (r.h) return r;
    // ONE RCCL per process: a Python host has already mapped PyTorch's bundled librccl (torch.distributed's "nccl"
    // backend); a second copy under the same SONAME would give the process two sets of communicator threads, proxy state and
    // environment parsing -- whatever the first multi-GPU run then does wrong would not be explained by either.  So: take the
    // copy that is ALREADY loaded if there is one (RTLD_NOLOAD), load one only otherwise, and record which file was bound.
    for (const char *name : {"librccl.so.1", "librccl.so"}) {
        r.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
        if (r.h) { r.was_loaded = true; break; }
    }
    if (!r.h)
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
    r.GetVersion = (decltype(r.GetVersion))dlsym(r.h, "ncclGetVersion");
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather && r.AllReduce;
    Dl_info di;
    if (r.AllGather && dladdr((void *)r.AllGather, &di) && di.dli_fname) r.path = di.dli_fname;
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

// Which RCCL this library bound: the file ncclAllGather was resolved from, its version code (ncclGetVersion: major * 10000 +
// minor * 100 + patch; 0 when the library has no such entry), and whether it had been loaded by somebody else before (1) or
// by this library (0).  path: at least path_len bytes, always terminated.  AMK_ERR_UNSUPPORTED when no librccl can be loaded.
int amk_shard_rccl_info(char *path, int path_len, int *version, int *was_loaded) {
    if (!path || path_len <= 0) return AMK_ERR_INVALID_ARG;
    path[0] = 0;
    Rccl &r = rccl();
    if (!r.ok) return AMK_ERR_UNSUPPORTED;
    std::strncpy(path, r.path.c_str(), (size_t)path_len - 1);
    path[path_len - 1] = 0;
    int v = 0;
    if (r.GetVersion && r.GetVersion(&v) != ncclSuccess) v = 0;
    if (version) *version = v;
    if (was_loaded) *was_loaded = r.was_loaded ? 1 : 0;
    return AMK_OK;
}

// Watchdog for the exchange step: waits until everything queued on `stream` so far (the gathers above included) has finished,
// polling hipStreamQuery, at most timeout_s seconds.  AMK_ERR_TIMEOUT: a collective is still not done -- a peer that never
// entered it, a dead link, two RCCL copies in one process ...; the communicator is then unusable and the process should exit
// (set NCCL_DEBUG=INFO and NCCL_DEBUG_SUBSYS=INIT,COLL for the next run: RCCL prints which rank / channel it waits for).
int amk_shard_wait(amk_shard *s, void *stream, double timeout_s) {
    if (!s || !(timeout_s > 0)) return AMK_ERR_INVALID_ARG;
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t e = hipStreamQuery((hipStream_t)stream);
        if (e == hipSuccess) return AMK_OK;
        if (e != hipErrorNotReady) return amk::hip_fail(e);
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) return AMK_ERR_TIMEOUT;
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 0.002)
            std::this_thread::sleep_for(std::chrono::microseconds(200));   // spin for the first 2 ms (a healthy gather), then doze
    }
}

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
