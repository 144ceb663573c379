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
#include <cstdlib>
#include <cstring>
#include <mutex>
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
    bool fake = false; // lab build: the stand-in was selected when the table was built (latched)
};

#ifdef VQA_TEST_SEAMS
// ---- test seam, lab build only (VQA_COMM_FAKE_RCCL=1): an in-library stand-in for the seven RCCL entry points, so that the
// SINGLE-PROCESS MULTI-CONTEXT path (device list, one scratch buffer per context on its own device, staging copies,
// group start/end bracketing, row layout of vals) can be exercised on a box with one GPU, where RCCL itself refuses
// two ranks on one device.  The fake checks the bracketing, sums on the host, and records every call in a trace
// (vqa_comm_debug_trace).  It is never selected unless the environment variable is set when the table is built, and
// the decision is LATCHED with the table (rccl_api::fake): the mode cannot flip once real RCCL has been loaded.  The
// stand-in's globals sit behind one mutex (the table itself was already thread-safe).
struct fake_comm { int rank, nranks, dev; };
struct fake_pending { const void *send; void *recv; size_t count; fake_comm *comm; hipStream_t st; };
static std::mutex g_fake_mu;
static std::vector<fake_pending> g_fake_pending;
static bool g_fake_in_group = false;
static std::string g_fake_trace;

static ncclResult_t fake_GetUniqueId(ncclUniqueId *u) { std::lock_guard<std::mutex> lk(g_fake_mu); memset(u, 0x5a, sizeof *u); g_fake_trace += "GetUniqueId;"; return ncclSuccess; }
static ncclResult_t fake_CommInitAll(ncclComm_t *comms, int n, const int *devs)
{
    std::lock_guard<std::mutex> lk(g_fake_mu);
    g_fake_trace += "CommInitAll(n=" + std::to_string(n) + ",devs=";
    for (int i = 0; i < n; i++) {
        comms[i] = (ncclComm_t) new fake_comm{i, n, devs[i]};
        g_fake_trace += std::to_string(devs[i]) + (i + 1 < n ? "," : "");
    }
    g_fake_trace += ");";
    return ncclSuccess;
}
static ncclResult_t fake_CommInitRank(ncclComm_t *comm, int n, ncclUniqueId, int rank)
{
    std::lock_guard<std::mutex> lk(g_fake_mu);
    if (n != 1) return ncclInvalidUsage; // the fake has no second process to meet
    *comm = (ncclComm_t) new fake_comm{rank, n, -1};
    g_fake_trace += "CommInitRank(n=1);";
    return ncclSuccess;
}
static ncclResult_t fake_CommDestroy(ncclComm_t c) { std::lock_guard<std::mutex> lk(g_fake_mu); delete (fake_comm *)c; g_fake_trace += "CommDestroy;"; return ncclSuccess; }
static ncclResult_t fake_GroupStart()
{
    std::lock_guard<std::mutex> lk(g_fake_mu);
    if (g_fake_in_group) return ncclInvalidUsage;
    g_fake_in_group = true;
    g_fake_pending.clear();
    g_fake_trace += "GroupStart;";
    return ncclSuccess;
}
static ncclResult_t fake_AllReduce(const void *send, void *recv, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t comm,
                                   hipStream_t st)
{
    std::lock_guard<std::mutex> lk(g_fake_mu);
    if (!g_fake_in_group || dt != ncclFloat64 || op != ncclSum) return ncclInvalidUsage; // must sit inside a group
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, recv) != hipSuccess) return ncclInvalidArgument;
    fake_comm *fc = (fake_comm *)comm;
    g_fake_trace += "AllReduce(rank=" + std::to_string(fc->rank) + ",count=" + std::to_string(count) + ",buf_dev=" +
                    std::to_string(at.device) + ");";
    g_fake_pending.push_back({send, recv, count, fc, st});
    return ncclSuccess;
}
static ncclResult_t fake_GroupEnd()
{
    std::lock_guard<std::mutex> lk(g_fake_mu);
    if (!g_fake_in_group) return ncclInvalidUsage;
    g_fake_in_group = false;
    g_fake_trace += "GroupEnd(" + std::to_string(g_fake_pending.size()) + ");";
    if (g_fake_pending.empty()) return ncclSuccess;
    if ((int)g_fake_pending.size() != g_fake_pending[0].comm->nranks) return ncclInvalidUsage; // every rank must take part
    const size_t count = g_fake_pending[0].count;
    std::vector<double> sum(count, 0.0), tmp(count);
    int dev0 = 0;
    (void)hipGetDevice(&dev0);
    for (auto &p : g_fake_pending) {
        if (p.comm->dev >= 0) (void)hipSetDevice(p.comm->dev); // (each context's buffer lives on its own device)
        if (p.count != count || hipStreamSynchronize(p.st) != hipSuccess) return ncclInvalidUsage;
        if (hipMemcpy(tmp.data(), p.send, sizeof(double) * count, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
        for (size_t i = 0; i < count; i++) sum[i] += tmp[i];
    }
    for (auto &p : g_fake_pending) {
        if (p.comm->dev >= 0) (void)hipSetDevice(p.comm->dev);
        if (hipMemcpy(p.recv, sum.data(), sizeof(double) * count, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    }
    (void)hipSetDevice(dev0);
    return ncclSuccess;
}
static const char *fake_GetErrorString(ncclResult_t r) { return r == ncclSuccess ? "ok" : "fake rccl: invalid usage"; }

static bool fake_env()
{
    const char *e = getenv("VQA_COMM_FAKE_RCCL");
    return e && e[0] == '1';
}
#endif // VQA_TEST_SEAMS

// Thread-safe by construction: a function-local static initialised once by the lambda (C++11 magic statics), so two
// host threads creating communicators at the same time can never see a half-filled table.
rccl_api &rccl()
{
    static rccl_api a = [] {
        rccl_api t;
#ifdef VQA_TEST_SEAMS
        if (fake_env()) {
            t.GetUniqueId = fake_GetUniqueId; t.CommInitAll = fake_CommInitAll; t.CommInitRank = fake_CommInitRank;
            t.CommDestroy = fake_CommDestroy; t.AllReduce = fake_AllReduce; t.GroupStart = fake_GroupStart;
            t.GroupEnd = fake_GroupEnd; t.GetErrorString = fake_GetErrorString;
            t.ok = t.fake = true;
            return t;
        }
#endif
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            t.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (t.h) break;
        }
        if (!t.h) return t;
#define SYM(field, name) t.field = (decltype(t.field))dlsym(t.h, name)
        SYM(GetUniqueId, "ncclGetUniqueId");
        SYM(CommInitAll, "ncclCommInitAll");
        SYM(CommInitRank, "ncclCommInitRank");
        SYM(CommDestroy, "ncclCommDestroy");
        SYM(AllReduce, "ncclAllReduce");
        SYM(GroupStart, "ncclGroupStart");
        SYM(GroupEnd, "ncclGroupEnd");
        SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
        t.ok = t.GetUniqueId && t.CommInitAll && t.CommInitRank && t.CommDestroy && t.AllReduce && t.GroupStart && t.GroupEnd;
        return t;
    }();
    return a;
}

// why the last vqa_comm_create / vqa_comm_create_rank of this thread failed (the communicator object does not
// survive a failed creation, so its own last_err cannot say): read with vqa_comm_last_error(NULL)
thread_local std::string g_create_err;
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
        for (int j = 0; j < i; j++) // one ctx per device (the test seam alone may put two on one: it never calls RCCL)
            if (vqa_ctx_device_(ctxs[i]) == vqa_ctx_device_(ctxs[j]) && !rccl().fake) {
                g_create_err = "vqa_comm_create: two contexts on device " + std::to_string(vqa_ctx_device_(ctxs[i]));
                return VQA_ERR_INVALID;
            }
    }
    if (!rccl().ok) { g_create_err = "librccl.so.1 could not be loaded"; return VQA_ERR_UNSUPPORTED; }
    vqa_comm *c = new vqa_comm;
    c->ctxs.assign(ctxs, ctxs + n_ctx);
    c->comms.resize(n_ctx);
    c->nranks = n_ctx;
    std::vector<int> devs(n_ctx);
    for (int i = 0; i < n_ctx; i++) devs[i] = vqa_ctx_device_(ctxs[i]);
    const ncclResult_t r = rccl().CommInitAll(c->comms.data(), n_ctx, devs.data());
    if (r != ncclSuccess) {
        g_create_err = std::string("ncclCommInitAll -> ") + (rccl().GetErrorString ? rccl().GetErrorString(r) : "rccl error");
        delete c;
        return VQA_ERR_HIP;
    }
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
    if (!rccl().ok) { g_create_err = "librccl.so.1 could not be loaded"; return VQA_ERR_UNSUPPORTED; }
    if (hipSetDevice(vqa_ctx_device_(ctx)) != hipSuccess) return VQA_ERR_HIP;
    vqa_comm *c = new vqa_comm;
    c->ctxs.push_back(ctx);
    c->comms.resize(1);
    c->nranks = n_ranks;
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    const ncclResult_t r = rccl().CommInitRank(&c->comms[0], n_ranks, u, rank);
    if (r != ncclSuccess) {
        g_create_err = std::string("ncclCommInitRank -> ") + (rccl().GetErrorString ? rccl().GetErrorString(r) : "rccl error");
        delete c;
        return VQA_ERR_HIP;
    }
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

const char *vqa_comm_last_error(const vqa_comm *c) { return c ? c->last_err.c_str() : g_create_err.c_str(); }

// the calls the test seam received, in order ("" unless VQA_COMM_FAKE_RCCL=1)
const char *vqa_comm_debug_trace(void)
{
#ifdef VQA_TEST_SEAMS
    return g_fake_trace.c_str();
#else
    return ""; // the shipped library has no stand-in
#endif
}

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
