// lh_comm_*: the gradient all-reduce of the data-parallel path at the C ABI.
//
// A thin layer over RCCL (one communicator per process = per GPU; ring / tree / direct algorithms over xGMI are RCCL's):
// init from a 128-byte unique id that rank 0 creates and the host side distributes, in-place sum all-reduce of one
// gradient bucket on the caller's stream (asynchronous; legal inside hipGraph capture), destroy.  The library does NOT
// link RCCL: it binds to the librccl the process has already loaded (PyTorch-ROCm brings its own), or loads the system
// one, so that a process never holds two RCCL instances.
#include "common.h"

#include <dlfcn.h>
#include <mutex>
#include <rccl/rccl.h>
#include <string.h>

namespace {
struct RcclApi {
    decltype(&ncclGetUniqueId) get_unique_id = nullptr;
    decltype(&ncclCommInitRank) comm_init_rank = nullptr;
    decltype(&ncclAllReduce) all_reduce = nullptr;
    decltype(&ncclReduceScatter) reduce_scatter = nullptr;
    decltype(&ncclAllGather) all_gather = nullptr;
    decltype(&ncclSend) send = nullptr;
    decltype(&ncclRecv) recv = nullptr;
    decltype(&ncclGroupStart) group_start = nullptr;
    decltype(&ncclGroupEnd) group_end = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
    bool ok = false;
};

const RcclApi& rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        void* h = nullptr;
        const char* names[] = {"librccl.so", "librccl.so.1"};
        for (const char* n : names)                      // already in the process (torch's copy)?
            if (!h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        for (const char* n : names)
            if (!h) h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (!h) return;
        api.get_unique_id = (decltype(api.get_unique_id))dlsym(h, "ncclGetUniqueId");
        api.comm_init_rank = (decltype(api.comm_init_rank))dlsym(h, "ncclCommInitRank");
        api.all_reduce = (decltype(api.all_reduce))dlsym(h, "ncclAllReduce");
        api.comm_destroy = (decltype(api.comm_destroy))dlsym(h, "ncclCommDestroy");
        api.reduce_scatter = (decltype(api.reduce_scatter))dlsym(h, "ncclReduceScatter");
        api.all_gather = (decltype(api.all_gather))dlsym(h, "ncclAllGather");
        api.send = (decltype(api.send))dlsym(h, "ncclSend");
        api.recv = (decltype(api.recv))dlsym(h, "ncclRecv");
        api.group_start = (decltype(api.group_start))dlsym(h, "ncclGroupStart");
        api.group_end = (decltype(api.group_end))dlsym(h, "ncclGroupEnd");
        api.error_string = (decltype(api.error_string))dlsym(h, "ncclGetErrorString");
        api.ok = api.get_unique_id && api.comm_init_rank && api.all_reduce && api.comm_destroy && api.error_string &&
                 api.reduce_scatter && api.all_gather && api.send && api.recv && api.group_start && api.group_end;
    });
    return api;
}

int fail(const char* what, ncclResult_t r) {
    lh_set_error("%s: %s", what, rccl().error_string ? rccl().error_string(r) : "RCCL error");
    return LH_ERR_HIP;
}
}  // namespace

struct lh_comm {
    ncclComm_t comm;
    int rank, nranks;
};

extern "C" int lh_comm_unique_id(void* id128) {
    LH_REQUIRE(id128, "lh_comm_unique_id: null pointer");
    if (!rccl().ok) { lh_set_error("lh_comm: librccl not found in this process"); return LH_ERR_UNSUPPORTED; }
    ncclUniqueId id;
    const ncclResult_t r = rccl().get_unique_id(&id);
    if (r != ncclSuccess) return fail("ncclGetUniqueId", r);
    static_assert(sizeof(id) == 128, "unique id size");
    memcpy(id128, &id, sizeof(id));
    return LH_OK;
}

extern "C" int lh_comm_init(lh_comm** comm, int rank, int nranks, const void* id128) {
    LH_REQUIRE(comm && id128 && nranks >= 1 && rank >= 0 && rank < nranks, "lh_comm_init: bad arguments");
    if (!rccl().ok) { lh_set_error("lh_comm: librccl not found in this process"); return LH_ERR_UNSUPPORTED; }
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    lh_comm* c = new lh_comm{nullptr, rank, nranks};
    const ncclResult_t r = rccl().comm_init_rank(&c->comm, nranks, id, rank);     // binds to the current HIP device
    if (r != ncclSuccess) { delete c; return fail("ncclCommInitRank", r); }
    *comm = c;
    return LH_OK;
}

extern "C" int lh_comm_allreduce_sum(lh_comm* comm, void* buf, size_t count, int dtype, void* stream) {
    LH_REQUIRE(comm && buf, "lh_comm_allreduce_sum: null pointer");
    ncclDataType_t t;
    switch (dtype) {
        case LH_F32: t = ncclFloat32; break;
        case LH_BF16: t = ncclBfloat16; break;
        case LH_F16: t = ncclFloat16; break;
        default: lh_set_error("lh_comm_allreduce_sum: bad dtype %d", dtype); return LH_ERR_ARG;
    }
    const ncclResult_t r = rccl().all_reduce(buf, buf, count, t, ncclSum, comm->comm, (hipStream_t)stream);
    if (r != ncclSuccess) return fail("ncclAllReduce", r);
    return LH_OK;
}

static int comm_dtype(const char* who, int dtype, ncclDataType_t* t, size_t* es) {
    switch (dtype) {
        case LH_F32: *t = ncclFloat32; *es = 4; return LH_OK;
        case LH_BF16: *t = ncclBfloat16; *es = 2; return LH_OK;
        case LH_F16: *t = ncclFloat16; *es = 2; return LH_OK;
    }
    lh_set_error("%s: bad dtype %d", who, dtype);
    return LH_ERR_ARG;
}

// recv[0 .. count) = sum over the ranks r of THEIR send[rank * count .. (rank + 1) * count): RCCL's reduce-scatter (its own algorithm).
extern "C" int lh_comm_reduce_scatter_sum(lh_comm* comm, const void* send, void* recv, size_t count, int dtype, void* stream) {
    LH_REQUIRE(comm && send && recv, "lh_comm_reduce_scatter_sum: null pointer");
    ncclDataType_t t; size_t es;
    if (comm_dtype("lh_comm_reduce_scatter_sum", dtype, &t, &es)) return LH_ERR_ARG;
    const ncclResult_t r = rccl().reduce_scatter(send, recv, count, t, ncclSum, comm->comm, (hipStream_t)stream);
    if (r != ncclSuccess) return fail("ncclReduceScatter", r);
    return LH_OK;
}

// recv[r * count .. (r + 1) * count) = rank r's send[0 .. count); recv + rank * count == send is the in-place form.
extern "C" int lh_comm_allgather(lh_comm* comm, const void* send, void* recv, size_t count, int dtype, void* stream) {
    LH_REQUIRE(comm && send && recv, "lh_comm_allgather: null pointer");
    ncclDataType_t t; size_t es;
    if (comm_dtype("lh_comm_allgather", dtype, &t, &es)) return LH_ERR_ARG;
    const ncclResult_t r = rccl().all_gather(send, recv, count, t, comm->comm, (hipStream_t)stream);
    if (r != ncclSuccess) return fail("ncclAllGather", r);
    return LH_OK;
}

// recv[r * count .. (r + 1) * count) = rank r's send[rank * count .. (rank + 1) * count): every rank exchanges one chunk with every
// other rank AT ONCE (grouped point-to-point sends / receives: over a fully connected xGMI mesh all seven links of a GPU carry a
// chunk each, 1/N of the buffer per link) -- the first half of the direct gradient exchange; send != recv.
extern "C" int lh_comm_alltoall(lh_comm* comm, const void* send, void* recv, size_t count, int dtype, void* stream) {
    LH_REQUIRE(comm && send && recv && send != recv, "lh_comm_alltoall: null pointer / in-place call");
    ncclDataType_t t; size_t es;
    if (comm_dtype("lh_comm_alltoall", dtype, &t, &es)) return LH_ERR_ARG;
    const RcclApi& a = rccl();
    // this rank's own chunk never touches the wire: a device copy on the same stream (a one-rank communicator exchanges nothing)
    if (hipMemcpyAsync((char*)recv + (size_t)comm->rank * count * es, (const char*)send + (size_t)comm->rank * count * es, count * es,
                       hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) {
        lh_set_error("lh_comm_alltoall: device copy of the own chunk failed");
        return LH_ERR_HIP;
    }
    if (comm->nranks == 1) return LH_OK;
    ncclResult_t r = a.group_start();
    if (r != ncclSuccess) return fail("ncclGroupStart", r);
    for (int p = 0; p < comm->nranks && r == ncclSuccess; ++p) {
        if (p == comm->rank) continue;
        r = a.send((const char*)send + (size_t)p * count * es, count, t, p, comm->comm, (hipStream_t)stream);
        if (r == ncclSuccess) r = a.recv((char*)recv + (size_t)p * count * es, count, t, p, comm->comm, (hipStream_t)stream);
    }
    const ncclResult_t e = a.group_end();
    if (r != ncclSuccess) return fail("ncclSend / ncclRecv", r);
    if (e != ncclSuccess) return fail("ncclGroupEnd", e);
    return LH_OK;
}

extern "C" int lh_comm_size(const lh_comm* comm, int* rank, int* nranks) {
    LH_REQUIRE(comm && rank && nranks, "lh_comm_size: null pointer");
    *rank = comm->rank; *nranks = comm->nranks;
    return LH_OK;
}

extern "C" int lh_comm_destroy(lh_comm* comm) {
    if (!comm) return LH_OK;
    const ncclResult_t r = rccl().comm_destroy(comm->comm);
    delete comm;
    if (r != ncclSuccess) return fail("ncclCommDestroy", r);
    return LH_OK;
}
