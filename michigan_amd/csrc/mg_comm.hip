// mg_comm.hip -- the C ABI's collectives (include/michigan_hip.h, group iv): thin RCCL calls on the caller's stream.
//
// RCCL is NOT a link-time dependency of libmichigan_hip.so: the library is dlopen'ed by SONAME at the first mg_comm_* call.  Inside a
// PyTorch-ROCm process the SONAME is already mapped (torch/lib/librccl.so carries SONAME librccl.so.1), so the dynamic loader hands back
// torch's copy and both users share one RCCL runtime; a stand-alone C host gets /opt/rocm/lib/librccl.so.1 through the usual search path.
// Only the five entry points below are resolved; their prototypes are restated here from <rccl/rccl.h> (NCCL's public, stable C API:
// ncclUniqueId is 128 opaque bytes, ncclComm_t an opaque pointer, ncclFloat32 = 7, ncclFloat64 = 8, ncclSum = 0).
#include "mg_common.h"
#include <dlfcn.h>
#include <string.h>
#include <mutex>

namespace {
typedef struct { char internal[MG_COMM_ID_BYTES]; } nccl_unique_id;
typedef void* nccl_comm;
enum { NCCL_SUCCESS = 0, NCCL_FLOAT32 = 7, NCCL_FLOAT64 = 8, NCCL_SUM = 0 };

struct rccl_api {
    void* handle = nullptr;
    int (*GetUniqueId)(nccl_unique_id*) = nullptr;
    int (*CommInitRank)(nccl_comm*, int, nccl_unique_id, int) = nullptr;
    int (*CommDestroy)(nccl_comm) = nullptr;
    int (*CommCount)(nccl_comm, int*) = nullptr;
    int (*CommUserRank)(nccl_comm, int*) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, nccl_comm, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
};

rccl_api& rccl()
{
    static rccl_api api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (api.handle) break;
        }
        if (!api.handle) return;
#define MG_SYM(field, name) *(void**)(&api.field) = dlsym(api.handle, name)
        MG_SYM(GetUniqueId, "ncclGetUniqueId");
        MG_SYM(CommInitRank, "ncclCommInitRank");
        MG_SYM(CommDestroy, "ncclCommDestroy");
        MG_SYM(CommCount, "ncclCommCount");
        MG_SYM(CommUserRank, "ncclCommUserRank");
        MG_SYM(AllReduce, "ncclAllReduce");
        MG_SYM(GetErrorString, "ncclGetErrorString");
#undef MG_SYM
        api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.CommCount && api.CommUserRank && api.AllReduce;
    });
    return api;
}

int rccl_fail(const char* what, int rc)
{
    rccl_api& r = rccl();
    return mg_fail(MG_ERR_LAUNCH, "%s: RCCL error %d (%s)", what, rc, r.GetErrorString ? r.GetErrorString(rc) : "?");
}

#define MG_NEED_RCCL(name) do { if (!rccl().ok) return mg_fail(MG_ERR_UNSUPPORTED, "%s: librccl.so.1 could not be loaded (%s)", name, \
                                                                 rccl().handle ? "missing symbols" : dlerror() ? dlerror() : "not found"); } while (0)
}  // namespace

extern "C" int mg_comm_unique_id(void* id_out)
{
    MG_CHECK_ARG(id_out != nullptr, "mg_comm_unique_id: null id buffer");
    MG_NEED_RCCL("mg_comm_unique_id");
    nccl_unique_id id;
    const int rc = rccl().GetUniqueId(&id);
    if (rc != NCCL_SUCCESS) return rccl_fail("mg_comm_unique_id", rc);
    memcpy(id_out, id.internal, MG_COMM_ID_BYTES);
    return MG_OK;
}

extern "C" int mg_comm_init(const void* id, int32_t rank, int32_t world, int64_t* comm_out)
{
    MG_CHECK_ARG(id != nullptr && comm_out != nullptr, "mg_comm_init: null pointer");
    MG_CHECK_ARG(world >= 1 && rank >= 0 && rank < world, "mg_comm_init: bad rank %d of %d", rank, world);
    MG_NEED_RCCL("mg_comm_init");
    nccl_unique_id uid;
    memcpy(uid.internal, id, MG_COMM_ID_BYTES);
    nccl_comm comm = nullptr;
    const int rc = rccl().CommInitRank(&comm, world, uid, rank);
    if (rc != NCCL_SUCCESS) return rccl_fail("mg_comm_init", rc);
    *comm_out = (int64_t)(intptr_t)comm;
    return MG_OK;
}

extern "C" int mg_comm_destroy(int64_t comm)
{
    MG_CHECK_ARG(comm != 0, "mg_comm_destroy: null communicator");
    MG_NEED_RCCL("mg_comm_destroy");
    const int rc = rccl().CommDestroy((nccl_comm)(intptr_t)comm);
    return rc == NCCL_SUCCESS ? MG_OK : rccl_fail("mg_comm_destroy", rc);
}

extern "C" int mg_comm_world(int64_t comm, int32_t* rank_out, int32_t* world_out)
{
    MG_CHECK_ARG(comm != 0 && rank_out != nullptr && world_out != nullptr, "mg_comm_world: null argument");
    MG_NEED_RCCL("mg_comm_world");
    int r = 0, w = 0;
    int rc = rccl().CommUserRank((nccl_comm)(intptr_t)comm, &r);
    if (rc == NCCL_SUCCESS) rc = rccl().CommCount((nccl_comm)(intptr_t)comm, &w);
    if (rc != NCCL_SUCCESS) return rccl_fail("mg_comm_world", rc);
    *rank_out = r; *world_out = w;
    return MG_OK;
}

static int all_reduce_sum(const char* name, int64_t comm, void* buf, int64_t n, int type, void* stream)
{
    MG_CHECK_ARG(comm != 0, "%s: null communicator", name);
    MG_CHECK_ARG(buf != nullptr && n > 0, "%s: null buffer or n = %lld", name, (long long)n);
    MG_NEED_RCCL(name);
    const int rc = rccl().AllReduce(buf, buf, (size_t)n, type, NCCL_SUM, (nccl_comm)(intptr_t)comm, (hipStream_t)stream);
    return rc == NCCL_SUCCESS ? MG_OK : rccl_fail(name, rc);
}

extern "C" int mg_allreduce_stats(int64_t comm, void* sums, int64_t n, int32_t is_f64, void* stream)
{
    return all_reduce_sum("mg_allreduce_stats", comm, sums, n, is_f64 ? NCCL_FLOAT64 : NCCL_FLOAT32, stream);
}

extern "C" int mg_allreduce_grads(int64_t comm, float* bucket, int64_t n, void* stream)
{
    return all_reduce_sum("mg_allreduce_grads", comm, bucket, n, NCCL_FLOAT32, stream);
}
