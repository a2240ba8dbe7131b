// The one collective of the path as part of the C ABI: an all-gather of the generated token-id grids over RCCL.
// Inference shards by sample and the decode loop needs no communication (every reduction of MaskGit.generate is per sample,
// muse_maskgit_pytorch.py:561,576,580,603); each rank ends with ids int64 [b][n], all < codebook size <= 65536, and the gather moves them as
// int32 (32 KiB per rank at BASELINE configs[1]): latency-bound, one ncclAllGather on the caller's stream, ranks one per GPU over xGMI.
//
// RCCL is resolved at run time (dlsym in the process, then dlopen of librccl): the library has no link-time dependency on it, a host
// without RCCL still loads every other entry point, and inside a PyTorch process the RCCL that torch.distributed already loaded is the
// one used (one RCCL per process).
#include <dlfcn.h>
#include <new>
#include <string.h>

#include "common.h"
#include "muse_hip_internal.h"

namespace {

typedef struct { char internal[128]; } rcclUniqueId;      // NCCL_UNIQUE_ID_BYTES = 128 (rccl.h:40-43)
typedef void* rcclComm_t;
typedef int (*fn_get_unique_id)(rcclUniqueId*);
typedef int (*fn_comm_init_rank)(rcclComm_t*, int, rcclUniqueId, int);
typedef int (*fn_comm_destroy)(rcclComm_t);
typedef int (*fn_all_gather)(const void*, void*, size_t, int, rcclComm_t, hipStream_t);
typedef const char* (*fn_error_string)(int);
constexpr int RCCL_INT32 = 2;                              // ncclInt32 (rccl.h ncclDataType_t)

struct Rccl {
    fn_get_unique_id get_unique_id = nullptr;
    fn_comm_init_rank comm_init_rank = nullptr;
    fn_comm_destroy comm_destroy = nullptr;
    fn_all_gather all_gather = nullptr;
    fn_error_string error_string = nullptr;
    bool tried = false, ok = false;
};
Rccl g_rccl;

bool load_rccl() {
    if (g_rccl.tried) return g_rccl.ok;
    g_rccl.tried = true;
    void* h = RTLD_DEFAULT;
    if (!dlsym(h, "ncclAllGather")) {
        h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return false;
    }
    g_rccl.get_unique_id = (fn_get_unique_id)dlsym(h, "ncclGetUniqueId");
    g_rccl.comm_init_rank = (fn_comm_init_rank)dlsym(h, "ncclCommInitRank");
    g_rccl.comm_destroy = (fn_comm_destroy)dlsym(h, "ncclCommDestroy");
    g_rccl.all_gather = (fn_all_gather)dlsym(h, "ncclAllGather");
    g_rccl.error_string = (fn_error_string)dlsym(h, "ncclGetErrorString");
    g_rccl.ok = g_rccl.get_unique_id && g_rccl.comm_init_rank && g_rccl.comm_destroy && g_rccl.all_gather;
    return g_rccl.ok;
}

int rccl_error(int rc, const char* where) {
    char msg[256];
    snprintf(msg, sizeof(msg), "%s: RCCL error %d (%s)", where, rc, g_rccl.error_string ? g_rccl.error_string(rc) : "?");
    return mm_set_error(MM_ERR_HIP, msg);
}

__global__ void narrow_ids_kernel(const int64_t* __restrict__ in, int32_t* __restrict__ out, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = (int32_t)in[i];
}
__global__ void widen_ids_kernel(const int32_t* __restrict__ in, int64_t* __restrict__ out, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = (int64_t)in[i];
}

}  // namespace

struct mm_comm {
    rcclComm_t comm;
    int rank, world;
};

extern "C" {

int mm_comm_unique_id(void* id_out) {
    if (!id_out) return mm_set_error(MM_ERR_SHAPE, "comm_unique_id: NULL");
    if (!load_rccl()) return mm_set_error(MM_ERR_UNSUPPORTED, "comm: librccl not found in the process or on the library path");
    rcclUniqueId id;
    const int rc = g_rccl.get_unique_id(&id);
    if (rc) return rccl_error(rc, "ncclGetUniqueId");
    memcpy(id_out, &id, sizeof(id));
    return MM_OK;
}

int mm_comm_create(const void* unique_id, int rank, int world, mm_comm_t** out) {
    if (!unique_id || !out || world <= 0 || rank < 0 || rank >= world) return mm_set_error(MM_ERR_SHAPE, "comm_create: bad arguments");
    if (!load_rccl()) return mm_set_error(MM_ERR_UNSUPPORTED, "comm: librccl not found in the process or on the library path");
    rcclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    mm_comm* c = new (std::nothrow) mm_comm();
    if (!c) return mm_set_error(MM_ERR_HIP, "out of host memory");
    c->rank = rank; c->world = world; c->comm = nullptr;
    const int rc = g_rccl.comm_init_rank(&c->comm, world, id, rank);      // on the calling thread's current HIP device
    if (rc) { delete c; return rccl_error(rc, "ncclCommInitRank"); }
    *out = c;
    return MM_OK;
}

void mm_comm_destroy(mm_comm_t* comm) {
    if (!comm) return;
    if (comm->comm && g_rccl.comm_destroy) (void)g_rccl.comm_destroy(comm->comm);
    delete comm;
}

int mm_comm_world(const mm_comm_t* comm) { return comm ? comm->world : 0; }
int mm_comm_rank(const mm_comm_t* comm) { return comm ? comm->rank : -1; }

size_t mm_allgather_ids_workspace_bytes(const mm_comm_t* comm, int64_t count) {
    if (!comm || count <= 0) return 0;
    return (size_t)(count + (int64_t)comm->world * count) * 4 + 512;
}

int mm_allgather_ids(mm_comm_t* comm, mm_stream_t stream, const int64_t* ids, int64_t count, int64_t* out, void* workspace, size_t workspace_bytes) {
    if (!comm || !comm->comm) return mm_set_error(MM_ERR_SHAPE, "allgather_ids: comm is NULL");
    if (count <= 0) return MM_OK;
    if (!ids || !out || !workspace) return mm_set_error(MM_ERR_SHAPE, "allgather_ids: NULL pointer");
    if (workspace_bytes < mm_allgather_ids_workspace_bytes(comm, count)) return mm_set_error(MM_ERR_WORKSPACE, "allgather_ids: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    int32_t* send = reinterpret_cast<int32_t*>(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    int32_t* recv = send + ((count + 63) & ~(int64_t)63);
    const long total = (long)comm->world * count;
    int blocks = (int)((count + 255) / 256);
    hipLaunchKernelGGL(narrow_ids_kernel, dim3(blocks > 1024 ? 1024 : blocks), dim3(256), 0, s, ids, send, (long)count);
    int rc = mm_check_launch("narrow_ids_kernel");
    if (rc) return rc;
    rc = g_rccl.all_gather(send, recv, (size_t)count, RCCL_INT32, comm->comm, s);      // rank order, the caller's stream: no host synchronisation
    if (rc) return rccl_error(rc, "ncclAllGather");
    blocks = (int)((total + 255) / 256);
    hipLaunchKernelGGL(widen_ids_kernel, dim3(blocks > 1024 ? 1024 : blocks), dim3(256), 0, s, recv, out, total);
    return mm_check_launch("widen_ids_kernel");
}

}  // extern "C"
