// comm.cpp -- the per-evaluation exchange of a BCM sharded one process per GPU, on the library's OWN stream.
//
// The reference moves 1 (log-likelihood) or 3 (gradient) doubles per worker over TCP for every objective evaluation
// (cuda_scalingdist/cg_solver.cpp:72-213: the master collects them worker by worker).  Here expert k lives on rank
// k mod W (cg_solver.cpp:93), every rank evaluates its experts as one group of shared launches, and the rows
// {LL_k, g_k} of ALL experts reach every rank by ONE ncclAllGather -- enqueued on the stream the evaluation runs on,
// directly behind its last kernel, followed by the copy into pinned host memory; the host waits ONCE, for the whole
// sequence.  (Rounds 1-5 went through torch.distributed from Python: a host wait for the evaluation, a staging copy,
// the collective, a blocking copy back -- 45-80 us per evaluation measured at one rank, profiles/r06_rehearse_*.json,
// beside 0.68 ms of device time for the two 1500-row experts a rank of the 8-GPU si24000 run owns.)
//
// RCCL is opened at run time (dlopen librccl.so.1): libcugp.so itself has no link-time dependency on it, and inside a
// process that has torch loaded the handle is torch's own copy of the library (same SONAME).
#include <dlfcn.h>

#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>          // types and enums only

#include "../../include/cugp.h"
#include "group.h"

namespace {

struct Rccl {
    void* so = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};
Rccl g_rccl;
std::once_flag g_rccl_once;

const Rccl& rccl()
{
    std::call_once(g_rccl_once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            g_rccl.so = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (g_rccl.so) break;
        }
        if (!g_rccl.so) return;
        auto sym = [](const char* n) { return dlsym(g_rccl.so, n); };
        g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))sym("ncclGetUniqueId");
        g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))sym("ncclCommInitRank");
        g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))sym("ncclCommDestroy");
        g_rccl.AllGather = (decltype(g_rccl.AllGather))sym("ncclAllGather");
        g_rccl.AllReduce = (decltype(g_rccl.AllReduce))sym("ncclAllReduce");
        g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))sym("ncclGetErrorString");
        g_rccl.ok = g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.AllGather && g_rccl.AllReduce;
    });
    return g_rccl;
}

int nccl_fail(const char* what, ncclResult_t r)
{
    char buf[256];
    snprintf(buf, sizeof buf, "%s: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error");
    return cugp_internal_fail(CUGP_ERR_DEVICE, buf);
}

int hip_fail(const char* what, hipError_t e)
{
    char buf[256];
    snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    return cugp_internal_fail(e == hipErrorOutOfMemory ? CUGP_ERR_NOMEM : CUGP_ERR_DEVICE, buf);
}

}  // namespace

struct cugp_comm {
    ncclComm_t comm = nullptr;   // null: a world of one without a communicator (nothing to exchange)
    int rank = 0, world = 1, device = 0;
    double *dsend = nullptr, *drecv = nullptr;   // [per][4] this rank's rows, [world * per][4] everybody's
    double* hrecv = nullptr;                     // pinned copy of drecv
    int per = 0;
};

extern "C" {

int cugp_comm_unique_id(void* id, int bytes)
{
    if (!id || bytes != NCCL_UNIQUE_ID_BYTES) return CUGP_ERR_INVALID;
    const Rccl& R = rccl();
    if (!R.ok) return cugp_internal_fail(CUGP_ERR_NODEVICE, "librccl.so.1 could not be opened (dlopen)");
    ncclUniqueId u;
    const ncclResult_t r = R.GetUniqueId(&u);
    if (r != ncclSuccess) return nccl_fail("ncclGetUniqueId", r);
    memcpy(id, &u, sizeof u);
    return CUGP_OK;
}

int cugp_comm_create(const void* id, int bytes, int rank, int world, int device, cugp_comm** out)
{
    if (!out || world < 1 || rank < 0 || rank >= world) return CUGP_ERR_INVALID;
    if (world > 1 && !id) return CUGP_ERR_INVALID;
    if (id && bytes != NCCL_UNIQUE_ID_BYTES) return CUGP_ERR_INVALID;
    cugp_comm* c = new (std::nothrow) cugp_comm;
    if (!c) return CUGP_ERR_NOMEM;
    c->rank = rank; c->world = world; c->device = device;
    if (id) {                                         // (a world of one WITH an id: a one-rank communicator, to rehearse the path)
        const Rccl& R = rccl();
        if (!R.ok) { delete c; return cugp_internal_fail(CUGP_ERR_NODEVICE, "librccl.so.1 could not be opened (dlopen)"); }
        hipError_t e = hipSetDevice(device);
        if (e != hipSuccess) { delete c; return hip_fail("hipSetDevice", e); }
        ncclUniqueId u;
        memcpy(&u, id, sizeof u);
        const ncclResult_t r = R.CommInitRank(&c->comm, world, u, rank);
        if (r != ncclSuccess) { delete c; return nccl_fail("ncclCommInitRank", r); }
    }
    *out = c;
    return CUGP_OK;
}

int cugp_comm_destroy(cugp_comm* c)
{
    if (!c) return CUGP_OK;
    (void)hipSetDevice(c->device);
    if (c->comm) (void)rccl().CommDestroy(c->comm);
    if (c->dsend) (void)hipFree(c->dsend);
    if (c->drecv) (void)hipFree(c->drecv);
    if (c->hrecv) (void)hipHostFree(c->hrecv);
    delete c;
    return CUGP_OK;
}

static int comm_buffers(cugp_comm* c, int per)
{
    if (c->per >= per) return CUGP_OK;
    if (c->dsend) (void)hipFree(c->dsend);
    if (c->drecv) (void)hipFree(c->drecv);
    if (c->hrecv) (void)hipHostFree(c->hrecv);
    c->dsend = c->drecv = c->hrecv = nullptr;
    c->per = 0;
    hipError_t e = hipMalloc((void**)&c->dsend, (size_t)per * 4 * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&c->drecv, (size_t)c->world * per * 4 * sizeof(double));
    if (e == hipSuccess) e = hipHostMalloc((void**)&c->hrecv, (size_t)c->world * per * 4 * sizeof(double), hipHostMallocDefault);
    if (e == hipSuccess) e = hipMemset(c->dsend, 0, (size_t)per * 4 * sizeof(double));   // unused slots: exact zeros, for good
    if (e != hipSuccess) return hip_fail("exchange buffers", e);
    c->per = per;
    return CUGP_OK;
}

// One objective evaluation of a sharded BCM on this rank: evaluate the local experts (b; may be null on a rank that owns
// none), all-gather everybody's rows, -> rows_out[world * per][4]: rank r's i-th expert (global expert r + i * world)
// in row r * per + i, {LL, g0, g1, g2}; slots beyond a rank's experts are zero.  Everything between the first kernel of
// the evaluation and the pinned copy of the gathered rows is ONE in-order sequence on the evaluation's stream.
int cugp_bcm_loglik_grad_allgather(cugp_bcm* b, cugp_comm* c, int per, double* rows_out)
{
    if (!c || per <= 0 || !rows_out) return CUGP_ERR_INVALID;
    int nlocal = 0;
    if (b && cugp_bcm_num_experts(b, &nlocal)) return CUGP_ERR_INVALID;
    if (nlocal > per) return CUGP_ERR_INVALID;
    hipError_t e = hipSetDevice(c->device);
    if (e != hipSuccess) return hip_fail("hipSetDevice", e);
    int rc = comm_buffers(c, per);
    if (rc) return rc;
    hipStream_t s = nullptr;
    if (nlocal > 0) {
        void* sv = nullptr;
        if ((rc = cugp_bcm_enqueue_rows_packed(b, c->dsend, &sv))) return rc;   // rows packed behind the evaluation, on its stream
        s = (hipStream_t)sv;
    }
    const size_t nsend = (size_t)per * 4, nall = nsend * c->world;
    const double* src = c->drecv;
    if (c->comm) {
        const ncclResult_t r = rccl().AllGather(c->dsend, c->drecv, nsend, ncclDouble, c->comm, s);
        if (r != ncclSuccess) { if (nlocal > 0) (void)cugp_bcm_finish_rows(b); return nccl_fail("ncclAllGather", r); }
    } else {
        src = c->dsend;                                // a world of one: this rank's rows are all the rows
    }
    e = hipMemcpyAsync(c->hrecv, src, nall * sizeof(double), hipMemcpyDeviceToHost, s);
    if (e != hipSuccess) { if (nlocal > 0) (void)cugp_bcm_finish_rows(b); return hip_fail("hipMemcpyAsync (gathered rows)", e); }
    if (nlocal > 0) rc = cugp_bcm_finish_rows(b);      // waits for the stream: evaluation, collective and copy
    else {
        e = hipStreamSynchronize(s);
        if (e != hipSuccess) return hip_fail("hipStreamSynchronize", e);
    }
    if (rc) return rc;
    memcpy(rows_out, c->hrecv, nall * sizeof(double));
    return CUGP_OK;
}

}  // extern "C"
