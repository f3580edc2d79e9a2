// comm_rccl.hip -- the multi-process side of include/ssf.h: RCCL (over xGMI inside a node) bound at run time.
//
// independent units (WDM channels simulated separately, launch-power sweep points, Monte-Carlo
// realisations) shard over the GPUs of a node, one process per GPU, with NO per-step communication; what crosses
// the links is the parameter block (ncclBroadcast), the inputs of the units a rank owns (ncclSend / ncclRecv from
// the root) and the results (ncclAllGather / send to the root).  librccl.so is opened with dlopen the first time a
// communicator is asked for, so single-GPU users never load it (it is large) and the library has no link-time
// dependency on it.  No reference equivalent (the reference is single-GPU, optic/models/modelsGPU.py).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>

#include "ssf_internal.h"

namespace {

struct Api {
    void *so = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    std::string err;
};

Api &api() {
    static Api a;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names)
            if ((a.so = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
        if (!a.so) {
            a.err = std::string("cannot open librccl.so: ") + dlerror();
            return;
        }
        bool ok = true;
        auto sym = [&](const char *n) {
            void *p = dlsym(a.so, n);
            if (!p) {
                ok = false;
                a.err = std::string("librccl.so lacks ") + n;
            }
            return p;
        };
        a.GetUniqueId = (decltype(a.GetUniqueId))sym("ncclGetUniqueId");
        a.CommInitRank = (decltype(a.CommInitRank))sym("ncclCommInitRank");
        a.CommDestroy = (decltype(a.CommDestroy))sym("ncclCommDestroy");
        a.GetErrorString = (decltype(a.GetErrorString))sym("ncclGetErrorString");
        a.AllReduce = (decltype(a.AllReduce))sym("ncclAllReduce");
        a.Broadcast = (decltype(a.Broadcast))sym("ncclBroadcast");
        a.AllGather = (decltype(a.AllGather))sym("ncclAllGather");
        a.Send = (decltype(a.Send))sym("ncclSend");
        a.Recv = (decltype(a.Recv))sym("ncclRecv");
        a.GroupStart = (decltype(a.GroupStart))sym("ncclGroupStart");
        a.GroupEnd = (decltype(a.GroupEnd))sym("ncclGroupEnd");
        if (!ok) {
            dlclose(a.so);
            a.so = nullptr;
        }
    });
    return a;
}

thread_local std::string g_comm_err;

}  // namespace

struct ssf_comm {
    int device = 0, nranks = 1, rank = 0;
    ncclComm_t comm = nullptr;
    hipStream_t st = nullptr;
    void *scratch[2] = {nullptr, nullptr};     // device staging for host buffers (send side / receive side)
    size_t cap[2] = {0, 0};
    std::string err;

    int fail(int code, const std::string &m) {
        err = m;
        return code;
    }
    int nccl(ncclResult_t r, const char *what) {
        if (r == ncclSuccess) return SSF_OK;
        return fail(SSF_ERR_COMM, std::string(what) + ": " + api().GetErrorString(r));
    }
    int hip(hipError_t e, const char *what) {
        if (e == hipSuccess) return SSF_OK;
        return fail(e == hipErrorOutOfMemory ? SSF_ERR_OOM : SSF_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
    }
    // a device pointer for `p`: p itself, or staging buffer `slot` (filled from p when `in`)
    int stage(const void *p, size_t bytes, int slot, bool in, void **dev) {
        if (ssf::on_device(p)) {
            *dev = const_cast<void *>(p);
            return SSF_OK;
        }
        if (cap[slot] < bytes) {
            if (scratch[slot]) (void)hipFree(scratch[slot]);
            scratch[slot] = nullptr;
            cap[slot] = 0;
            if (int rc = hip(hipMalloc(&scratch[slot], bytes), "hipMalloc(staging)")) return rc;
            cap[slot] = bytes;
        }
        *dev = scratch[slot];
        if (in)
            if (int rc = hip(hipMemcpyAsync(scratch[slot], p, bytes, hipMemcpyHostToDevice, st), "stage in")) return rc;
        return SSF_OK;
    }
    int unstage(void *p, const void *dev, size_t bytes) {
        if (dev != p)
            if (int rc = hip(hipMemcpyAsync(p, dev, bytes, hipMemcpyDeviceToHost, st), "stage out")) return rc;
        return hip(hipStreamSynchronize(st), "hipStreamSynchronize");
    }
};

extern "C" {

const char *ssf_comm_last_error(const ssf_comm *c) { return c ? c->err.c_str() : g_comm_err.c_str(); }

int ssf_comm_get_id(void *id) {
    if (!id) return SSF_ERR_BAD_ARG;
    Api &a = api();
    if (!a.so) {
        g_comm_err = a.err;
        return SSF_ERR_COMM;
    }
    ncclUniqueId u;
    ncclResult_t r = a.GetUniqueId(&u);
    if (r != ncclSuccess) {
        g_comm_err = std::string("ncclGetUniqueId: ") + a.GetErrorString(r);
        return SSF_ERR_COMM;
    }
    static_assert(sizeof(u) == SSF_COMM_ID_BYTES, "ncclUniqueId size");
    std::memcpy(id, &u, sizeof(u));
    return SSF_OK;
}

int ssf_comm_create(int device, int32_t nranks, int32_t rank, const void *id, ssf_comm **out) {
    if (!out || !id || nranks < 1 || rank < 0 || rank >= nranks) {
        g_comm_err = "ssf_comm_create: bad argument";
        return SSF_ERR_BAD_ARG;
    }
    *out = nullptr;
    Api &a = api();
    if (!a.so) {
        g_comm_err = a.err;
        return SSF_ERR_COMM;
    }
    if (hipSetDevice(device) != hipSuccess) {
        g_comm_err = "hipSetDevice failed";
        return SSF_ERR_NO_DEVICE;
    }
    auto *c = new ssf_comm();
    c->device = device;
    c->nranks = nranks;
    c->rank = rank;
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof(u));
    int rc = c->hip(hipStreamCreateWithFlags(&c->st, hipStreamNonBlocking), "hipStreamCreate");
    if (!rc) rc = c->nccl(a.CommInitRank(&c->comm, nranks, u, rank), "ncclCommInitRank");
    if (rc) {
        g_comm_err = c->err;
        if (c->st) (void)hipStreamDestroy(c->st);
        delete c;
        return rc;
    }
    *out = c;
    return SSF_OK;
}

int ssf_comm_destroy(ssf_comm *c) {
    if (!c) return SSF_OK;
    (void)hipSetDevice(c->device);
    if (c->comm) (void)api().CommDestroy(c->comm);
    for (void *p : c->scratch)
        if (p) (void)hipFree(p);
    if (c->st) (void)hipStreamDestroy(c->st);
    delete c;
    return SSF_OK;
}

int ssf_comm_allreduce(ssf_comm *c, double *values, int32_t n, int32_t op) {
    if (!c || !values || n < 1 || op < 0 || op > 1) return SSF_ERR_BAD_ARG;
    if (int rc = c->hip(hipSetDevice(c->device), "hipSetDevice")) return rc;
    void *d = nullptr;
    const size_t bytes = sizeof(double) * (size_t)n;
    if (int rc = c->stage(values, bytes, 0, true, &d)) return rc;
    if (int rc = c->nccl(api().AllReduce(d, d, (size_t)n, ncclFloat64, op == 0 ? ncclSum : ncclMax, c->comm, c->st), "ncclAllReduce")) return rc;
    return c->unstage(values, d, bytes);
}

int ssf_comm_barrier(ssf_comm *c) {
    double one = 1.0;
    return ssf_comm_allreduce(c, &one, 1, 0);
}

int ssf_comm_bcast(ssf_comm *c, void *buf, int64_t bytes, int32_t root) {
    if (!c || !buf || bytes < 1 || root < 0 || root >= c->nranks) return SSF_ERR_BAD_ARG;
    if (int rc = c->hip(hipSetDevice(c->device), "hipSetDevice")) return rc;
    void *d = nullptr;
    if (int rc = c->stage(buf, (size_t)bytes, 0, c->rank == root, &d)) return rc;
    if (int rc = c->nccl(api().Broadcast(d, d, (size_t)bytes, ncclUint8, root, c->comm, c->st), "ncclBroadcast")) return rc;
    return c->unstage(buf, d, (size_t)bytes);
}

int ssf_comm_send(ssf_comm *c, const void *buf, int64_t bytes, int32_t peer) {
    if (!c || !buf || bytes < 1 || peer < 0 || peer >= c->nranks || peer == c->rank) return SSF_ERR_BAD_ARG;
    if (int rc = c->hip(hipSetDevice(c->device), "hipSetDevice")) return rc;
    void *d = nullptr;
    if (int rc = c->stage(buf, (size_t)bytes, 0, true, &d)) return rc;
    if (int rc = c->nccl(api().Send(d, (size_t)bytes, ncclUint8, peer, c->comm, c->st), "ncclSend")) return rc;
    return c->hip(hipStreamSynchronize(c->st), "hipStreamSynchronize");
}

int ssf_comm_recv(ssf_comm *c, void *buf, int64_t bytes, int32_t peer) {
    if (!c || !buf || bytes < 1 || peer < 0 || peer >= c->nranks || peer == c->rank) return SSF_ERR_BAD_ARG;
    if (int rc = c->hip(hipSetDevice(c->device), "hipSetDevice")) return rc;
    void *d = nullptr;
    if (int rc = c->stage(buf, (size_t)bytes, 1, false, &d)) return rc;
    if (int rc = c->nccl(api().Recv(d, (size_t)bytes, ncclUint8, peer, c->comm, c->st), "ncclRecv")) return rc;
    return c->unstage(buf, d, (size_t)bytes);
}

int ssf_comm_allgather(ssf_comm *c, const void *send, void *recv, int64_t bytes_per_rank) {
    if (!c || !send || !recv || bytes_per_rank < 1) return SSF_ERR_BAD_ARG;
    if (int rc = c->hip(hipSetDevice(c->device), "hipSetDevice")) return rc;
    void *ds = nullptr, *dr = nullptr;
    const size_t b = (size_t)bytes_per_rank;
    if (int rc = c->stage(send, b, 0, true, &ds)) return rc;
    if (int rc = c->stage(recv, b * (size_t)c->nranks, 1, false, &dr)) return rc;
    if (int rc = c->nccl(api().AllGather(ds, dr, b, ncclUint8, c->comm, c->st), "ncclAllGather")) return rc;
    return c->unstage(recv, dr, b * (size_t)c->nranks);
}

int ssf_comm_rank(const ssf_comm *c) { return c ? c->rank : SSF_ERR_BAD_ARG; }
int ssf_comm_size(const ssf_comm *c) { return c ? c->nranks : SSF_ERR_BAD_ARG; }

}  // extern "C"

namespace ssf {
// all-gather of device buffers enqueued on the CALLER's stream, no synchronisation: the coupled fused engine puts it between the
// column launch that leaves its partial sums and the row launch that uses them (ssf_set_coupling_comm)
int comm_allgather_on(ssf_comm *c, const void *send_dev, void *recv_dev, size_t bytes_per_rank, hipStream_t st) {
    if (!c || !send_dev || !recv_dev || bytes_per_rank < 1) return SSF_ERR_BAD_ARG;
    return c->nccl(api().AllGather(send_dev, recv_dev, bytes_per_rank, ncclUint8, c->comm, st), "ncclAllGather");
}
int comm_nranks(const ssf_comm *c) { return c ? c->nranks : 0; }
int comm_device(const ssf_comm *c) { return c ? c->device : -1; }
const char *comm_error(const ssf_comm *c) { return c ? c->err.c_str() : ""; }
}  // namespace ssf
