// vqa_comm.hip — the ONE collective of the path, in the C ABI: a SUM all-reduce of a handful of pooled scalars
// over RCCL/xGMI.  The reference has no communication layer at all (its only parallelism is a
// ProcessPoolExecutor over frames, complexity_metrics.py:143); this is what a non-Python host uses after every
// device has pooled its own stream (or its shard of one stream, with the data-independent weights of
// pooling.py: mean(ewm(x)) = sum_i c_i x_i, so partial sums add up to the reference's pooled value).
//
// RCCL is loaded on first use (dlopen "librccl.so.1"): the library has no link-time dependency on it and a host
// that never creates a communicator never needs it.  Messages are <= 64 doubles: latency-bound, the xGMI link
// bandwidth is irrelevant, so there is nothing to tune here — correctness and clean failure are the whole job.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/vqa.h"

extern "C" int vqa_ctx_device_(const vqa_ctx *c);      // vqa_capi.hip
extern "C" void *vqa_ctx_stream_(const vqa_ctx *c);    // vqa_capi.hip

namespace {
struct rccl_api {
    void *h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

rccl_api &rccl()
{
    static rccl_api a;
    static bool tried = false;
    if (tried) return a;
    tried = true;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
        a.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (a.h) break;
    }
    if (!a.h) return a;
#define SYM(field, name) a.field = (decltype(a.field))dlsym(a.h, name)
    SYM(GetUniqueId, "ncclGetUniqueId");
    SYM(CommInitAll, "ncclCommInitAll");
    SYM(CommInitRank, "ncclCommInitRank");
    SYM(CommDestroy, "ncclCommDestroy");
    SYM(AllReduce, "ncclAllReduce");
    SYM(GroupStart, "ncclGroupStart");
    SYM(GroupEnd, "ncclGroupEnd");
    SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    a.ok = a.GetUniqueId && a.CommInitAll && a.CommInitRank && a.CommDestroy && a.AllReduce && a.GroupStart && a.GroupEnd;
    return a;
}
} // namespace

struct vqa_comm {
    std::vector<vqa_ctx *> ctxs;      // local contexts (one per local device)
    std::vector<ncclComm_t> comms;    // one communicator handle per local context
    std::vector<double *> dbuf;       // 64 doubles of device scratch per local context
    int nranks = 0;                   // ranks in the whole communicator
    std::string last_err;
};

#define VQA_COMM_MAX 64

static int fail(vqa_comm *c, const char *what, ncclResult_t r)
{
    if (c) {
        char b[256];
        snprintf(b, sizeof b, "%s -> %s", what, rccl().GetErrorString ? rccl().GetErrorString(r) : "rccl error");
        c->last_err = b;
    }
    return VQA_ERR_HIP;
}

static int alloc_scratch(vqa_comm *c)
{
    for (size_t i = 0; i < c->ctxs.size(); i++) {
        if (hipSetDevice(vqa_ctx_device_(c->ctxs[i])) != hipSuccess) return VQA_ERR_HIP;
        double *p = nullptr;
        if (hipMalloc(&p, sizeof(double) * VQA_COMM_MAX) != hipSuccess) return VQA_ERR_OOM;
        c->dbuf.push_back(p);
    }
    return VQA_OK;
}

extern "C" {

int vqa_comm_create(vqa_ctx *const *ctxs, int n_ctx, vqa_comm **out)
{
    if (!ctxs || n_ctx <= 0 || !out) return VQA_ERR_INVALID;
    for (int i = 0; i < n_ctx; i++) {
        if (!ctxs[i]) return VQA_ERR_INVALID;
        for (int j = 0; j < i; j++)
            if (vqa_ctx_device_(ctxs[i]) == vqa_ctx_device_(ctxs[j])) return VQA_ERR_INVALID; // one ctx per device
    }
    if (!rccl().ok) return VQA_ERR_UNSUPPORTED; // librccl.so.1 not installed
    vqa_comm *c = new vqa_comm;
    c->ctxs.assign(ctxs, ctxs + n_ctx);
    c->comms.resize(n_ctx);
    c->nranks = n_ctx;
    std::vector<int> devs(n_ctx);
    for (int i = 0; i < n_ctx; i++) devs[i] = vqa_ctx_device_(ctxs[i]);
    const ncclResult_t r = rccl().CommInitAll(c->comms.data(), n_ctx, devs.data());
    if (r != ncclSuccess) { delete c; return VQA_ERR_HIP; }
    const int rc = alloc_scratch(c);
    if (rc) { vqa_comm_destroy(c); return rc; }
    *out = c;
    return VQA_OK;
}

int vqa_comm_unique_id(void *id, size_t id_bytes)
{
    if (!id || id_bytes < VQA_COMM_ID_BYTES) return VQA_ERR_INVALID;
    if (!rccl().ok) return VQA_ERR_UNSUPPORTED;
    static_assert(VQA_COMM_ID_BYTES == sizeof(ncclUniqueId), "vqa.h: VQA_COMM_ID_BYTES");
    ncclUniqueId u;
    if (rccl().GetUniqueId(&u) != ncclSuccess) return VQA_ERR_HIP;
    memcpy(id, &u, sizeof u);
    return VQA_OK;
}

int vqa_comm_create_rank(vqa_ctx *ctx, const void *id, size_t id_bytes, int n_ranks, int rank, vqa_comm **out)
{
    if (!ctx || !id || id_bytes < VQA_COMM_ID_BYTES || n_ranks <= 0 || rank < 0 || rank >= n_ranks || !out) return VQA_ERR_INVALID;
    if (!rccl().ok) return VQA_ERR_UNSUPPORTED;
    if (hipSetDevice(vqa_ctx_device_(ctx)) != hipSuccess) return VQA_ERR_HIP;
    vqa_comm *c = new vqa_comm;
    c->ctxs.push_back(ctx);
    c->comms.resize(1);
    c->nranks = n_ranks;
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    const ncclResult_t r = rccl().CommInitRank(&c->comms[0], n_ranks, u, rank);
    if (r != ncclSuccess) { delete c; return VQA_ERR_HIP; }
    const int rc = alloc_scratch(c);
    if (rc) { vqa_comm_destroy(c); return rc; }
    *out = c;
    return VQA_OK;
}

int vqa_comm_destroy(vqa_comm *c)
{
    if (!c) return VQA_ERR_INVALID;
    for (size_t i = 0; i < c->dbuf.size(); i++) {
        (void)hipSetDevice(vqa_ctx_device_(c->ctxs[i]));
        (void)hipFree(c->dbuf[i]);
    }
    for (ncclComm_t k : c->comms)
        if (k) (void)rccl().CommDestroy(k);
    delete c;
    return VQA_OK;
}

int vqa_comm_size(const vqa_comm *c) { return c ? c->nranks : VQA_ERR_INVALID; }

const char *vqa_comm_last_error(const vqa_comm *c) { return c ? c->last_err.c_str() : ""; }

// vals: [local contexts][count] doubles, row i belongs to the i-th local context; in place.
int vqa_allreduce(vqa_comm *c, double *vals, int count)
{
    if (!c || !vals || count <= 0 || count > VQA_COMM_MAX) return VQA_ERR_INVALID;
    const size_t nl = c->ctxs.size();
    for (size_t i = 0; i < nl; i++) {
        if (hipSetDevice(vqa_ctx_device_(c->ctxs[i])) != hipSuccess) return VQA_ERR_HIP;
        if (hipMemcpyAsync(c->dbuf[i], vals + i * count, sizeof(double) * count, hipMemcpyHostToDevice,
                           (hipStream_t)vqa_ctx_stream_(c->ctxs[i])) != hipSuccess) return VQA_ERR_HIP;
    }
    ncclResult_t r = rccl().GroupStart();
    if (r != ncclSuccess) return fail(c, "ncclGroupStart", r);
    for (size_t i = 0; i < nl; i++) {
        r = rccl().AllReduce(c->dbuf[i], c->dbuf[i], (size_t)count, ncclFloat64, ncclSum, c->comms[i],
                             (hipStream_t)vqa_ctx_stream_(c->ctxs[i]));
        if (r != ncclSuccess) { (void)rccl().GroupEnd(); return fail(c, "ncclAllReduce", r); }
    }
    r = rccl().GroupEnd();
    if (r != ncclSuccess) return fail(c, "ncclGroupEnd", r);
    for (size_t i = 0; i < nl; i++) {
        if (hipSetDevice(vqa_ctx_device_(c->ctxs[i])) != hipSuccess) return VQA_ERR_HIP;
        hipStream_t st = (hipStream_t)vqa_ctx_stream_(c->ctxs[i]);
        if (hipMemcpyAsync(vals + i * count, c->dbuf[i], sizeof(double) * count, hipMemcpyDeviceToHost, st) != hipSuccess) return VQA_ERR_HIP;
        if (hipStreamSynchronize(st) != hipSuccess) return VQA_ERR_HIP;
    }
    return VQA_OK;
}

} // extern "C"
