// The step's one exchange: in-place sum over ranks of the packed [score | lik] buffer with RCCL, on the stream the
// kernels run on (no cross-stream event hop).  RCCL is bound lazily with dlopen so that libpacoh_gp.so has no link-time
// dependency on it and shares the librccl instance PyTorch already mapped, if any.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <mutex>

#include "pacoh_gp.h"

namespace {

struct RcclApi {
    ncclResult_t (*get_unique_id)(ncclUniqueId*) = nullptr;
    ncclResult_t (*comm_init_rank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*all_reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*comm_destroy)(ncclComm_t) = nullptr;
    bool ok = false;
};

RcclApi g_rccl;
std::once_flag g_rccl_once;

void bind_rccl() {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* nm : names)                       // a copy already in the process (PyTorch's) wins
        if ((h = dlopen(nm, RTLD_NOW | RTLD_NOLOAD))) break;
    if (!h)
        for (const char* nm : names)
            if ((h = dlopen(nm, RTLD_NOW | RTLD_LOCAL))) break;
    if (!h) return;
    g_rccl.get_unique_id = reinterpret_cast<decltype(g_rccl.get_unique_id)>(dlsym(h, "ncclGetUniqueId"));
    g_rccl.comm_init_rank = reinterpret_cast<decltype(g_rccl.comm_init_rank)>(dlsym(h, "ncclCommInitRank"));
    g_rccl.all_reduce = reinterpret_cast<decltype(g_rccl.all_reduce)>(dlsym(h, "ncclAllReduce"));
    g_rccl.comm_destroy = reinterpret_cast<decltype(g_rccl.comm_destroy)>(dlsym(h, "ncclCommDestroy"));
    g_rccl.ok = g_rccl.get_unique_id && g_rccl.comm_init_rank && g_rccl.all_reduce && g_rccl.comm_destroy;
}

const RcclApi* rccl() {
    std::call_once(g_rccl_once, bind_rccl);
    return g_rccl.ok ? &g_rccl : nullptr;
}

}  // namespace

extern "C" {

int pacoh_comm_unique_id(void* id_out) {
    static_assert(sizeof(ncclUniqueId) == PACOH_COMM_ID_BYTES, "ncclUniqueId size");
    if (!id_out) return PACOH_EINVAL;
    const RcclApi* api = rccl();
    if (!api) return PACOH_ENOCOMM;
    return static_cast<int>(api->get_unique_id(static_cast<ncclUniqueId*>(id_out)));
}

int pacoh_comm_init(const void* id, int rank, int world, void** comm_out) {
    if (!id || !comm_out || world < 1 || rank < 0 || rank >= world) return PACOH_EINVAL;
    const RcclApi* api = rccl();
    if (!api) return PACOH_ENOCOMM;
    ncclUniqueId uid;
    __builtin_memcpy(&uid, id, sizeof(uid));
    ncclComm_t comm = nullptr;
    ncclResult_t r = api->comm_init_rank(&comm, world, uid, rank);
    *comm_out = (r == ncclSuccess) ? static_cast<void*>(comm) : nullptr;
    return static_cast<int>(r);
}

int pacoh_allreduce_sum(void* buf, long count, int dtype, void* comm, void* stream) {
    if (!buf || !comm || count < 0) return PACOH_EINVAL;
    if (dtype != PACOH_F32 && dtype != PACOH_F64) return PACOH_EDTYPE;
    const RcclApi* api = rccl();
    if (!api) return PACOH_ENOCOMM;
    if (count == 0) return PACOH_OK;
    return static_cast<int>(api->all_reduce(buf, buf, static_cast<size_t>(count), dtype == PACOH_F32 ? ncclFloat32 : ncclFloat64,
                                            ncclSum, static_cast<ncclComm_t>(comm), static_cast<hipStream_t>(stream)));
}

int pacoh_comm_destroy(void* comm) {
    if (!comm) return PACOH_EINVAL;
    const RcclApi* api = rccl();
    if (!api) return PACOH_ENOCOMM;
    return static_cast<int>(api->comm_destroy(static_cast<ncclComm_t>(comm)));
}

}  // extern "C"
