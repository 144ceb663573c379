// vqa_capi.hip — the C ABI declared in include/vqa.h: context, memory, and the
// orchestration of the gfx950 kernels for one batch of frames.
//
// No CPU fallback lives here: every metric is produced by a HIP kernel or the
// call fails.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <tuple>
#include <vector>

#include "vqa_kernels.hpp"

using namespace vqa;

namespace {

struct dbuf {
    void *p = nullptr;
    size_t cap = 0;
};

struct resize_tabs {
    int32_t *xofs = nullptr, *xa = nullptr, *yofs = nullptr, *yb = nullptr;
    int mode = 0;
};

inline int align_up(int v, int a) { return (v + a - 1) / a * a; }

// one cached per-geometry table set and when it was last handed out (least recently used goes first, vqa.h "Memory")
template <class T> struct cached {
    T v;
    uint64_t used = 0;
};

} // namespace

struct vqa_ctx {
    int device = -1;
    hipStream_t stream = nullptr;
    hipStream_t side[3] = {nullptr, nullptr, nullptr};   // VQA_OPT_OVERLAP: block-SAD, the Canny chain and the full-frame DCT on their own streams
    hipEvent_t ev_fork = nullptr, ev_join[3] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_xctx = nullptr;   // vqa_stream_wait: "everything enqueued on this ctx so far" for another ctx's stream to wait on
    hipEvent_t fb_ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}; // Farneback: expansions of level k ready (0..3), planes ready (4), chunk done (5)
    std::string last_err;
    // options (vqa_set_option)
    bool opt_overlap = true, opt_hyst_stats = false;
    // -DVQA_TEST_SEAMS only (lab build; read from the environment once, in vqa_create)
    int seam_hyst_max_rounds = 0;   // VQA_HYST_MAX_ROUNDS: the hysteresis tail's round bound (0 = the shipped bound)
    int seam_hyst_rescue_max_rounds = 0; // VQA_HYST_RESCUE_MAX_ROUNDS: the rescue pass's bound (0 = the proof's: edge_weak + 2)
    long seam_fail_at = 0;          // VQA_FAIL_ENSURE_AT=N: the N-th scratch reservation of this ctx reports VQA_ERR_OOM
    long seam_ensure_calls = 0;
    long long seam_fb_chunk_bytes = 0; // VQA_FB_CHUNK_BYTES: Farneback's scratch budget per chunk (0 = the shipped 12 GiB): lets a small batch span chunks

    // device scratch (grow-only)
    dbuf gray_full, planeA, planeB, state, res_dev, partials, tile_flags, dirty0, dirty1, again_dev;
    dbuf stage_frames, stage_prev, dct_scratch, dct_pe, dct_pt;
    dbuf qres_dev, qpartials, qstage_ref, qstage_dist;
    // pinned host staging
    void *res_host = nullptr; size_t res_host_cap = 0;
    void *qres_host = nullptr; size_t qres_host_cap = 0;

    // per-geometry tables, at most VQA_TABLE_CACHE_GEOMETRIES of each kind (cache_put evicts the least recently used)
    std::map<std::tuple<int, int, int, int>, cached<resize_tabs>> tabs;
    std::map<std::tuple<int, int, int, int>, cached<fb_resize_tabs>> fb_tabs;
    std::map<int, cached<float *>> dct_mats;
    std::map<int, cached<dct_fft_plan>> fft_plans;  // k_dct_fft.hip: twiddle / post-twiddle tables per transform length
    uint64_t cache_clock = 0;
    int dct_wave_slots = 0;                 // wave slots of THIS device for the marching DCT's chunking (set in vqa_create)
    dbuf fb_tmp, fb_blur, fb_img, fb_R, fb_M, fb_flow0, fb_flow1, fb_part;

    // pending work
    int pend_c = 0, pend_q = 0;
    bool pend_c_prev0 = false, pend_c_tail_only = false;
    // geometry of the last complexity batch (debug reads)
    int last_n = 0, last_h = 0, last_w = 0, last_ph = 0, last_pw = 0, last_pp = 0, last_gp = 0;
    bool last_resized = false, last_has_full = false, last_has_state = false, last_has_planes = false;

    // per-kernel timing
    bool prof_on = false;
    std::vector<hipEvent_t> ev_pool;                   // recycled events
    std::vector<std::tuple<int, hipEvent_t, hipEvent_t>> ev_open; // (kernel id, start, stop) not yet read
    double prof_ms[VQA_K_COUNT] = {0};
    int64_t prof_n[VQA_K_COUNT] = {0};
};

namespace {
// RAII bracket: records start now and stop at scope exit when profiling is on.
struct prof_scope {
    vqa_ctx *c; int id; hipEvent_t a = nullptr, b = nullptr;
    static hipEvent_t get(vqa_ctx *c)
    {
        if (!c->ev_pool.empty()) { hipEvent_t e = c->ev_pool.back(); c->ev_pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        return e;
    }
    hipStream_t s;
    prof_scope(vqa_ctx *c_, int id_, hipStream_t s_ = nullptr) : c(c_), id(id_), s(s_ ? s_ : c_->stream)
    {
        if (!c->prof_on) return;
        a = get(c); b = get(c);
        if (a && b) (void)hipEventRecord(a, s);
    }
    ~prof_scope()
    {
        if (!a || !b) return;
        (void)hipEventRecord(b, s);
        c->ev_open.emplace_back(id, a, b);
    }
};

// after the stream has been synchronised: fold the open event pairs into the totals
void prof_collect(vqa_ctx *c)
{
    for (auto &t : c->ev_open) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, std::get<1>(t), std::get<2>(t)) == hipSuccess) {
            c->prof_ms[std::get<0>(t)] += ms;
            c->prof_n[std::get<0>(t)] += 1;
        }
        c->ev_pool.push_back(std::get<1>(t));
        c->ev_pool.push_back(std::get<2>(t));
    }
    c->ev_open.clear();
}

// a failed submit: its event pairs go back to the pool unread
void prof_discard(vqa_ctx *c)
{
    for (auto &t : c->ev_open) { c->ev_pool.push_back(std::get<1>(t)); c->ev_pool.push_back(std::get<2>(t)); }
    c->ev_open.clear();
}
} // namespace

#define HIPCHK(ctx, call)                                                                                  \
    do {                                                                                                   \
        hipError_t e_ = (call);                                                                            \
        if (e_ != hipSuccess) {                                                                            \
            char b_[512];                                                                                  \
            snprintf(b_, sizeof b_, "%s -> %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            (ctx)->last_err = b_;                                                                          \
            return e_ == hipErrorOutOfMemory ? VQA_ERR_OOM : VQA_ERR_HIP;                                  \
        }                                                                                                  \
    } while (0)

// every stream this ctx may have work on (the main one and, once created, the three side streams of VQA_OPT_OVERLAP)
static int sync_all(vqa_ctx *c)
{
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < 3; i++)
        if (c->side[i]) HIPCHK(c, hipStreamSynchronize(c->side[i]));
    return VQA_OK;
}

// lab build: VQA_FAIL_ENSURE_AT=N makes the N-th device reservation of this ctx (scratch buffer or table) report OOM
static inline bool seam_reservation_fails(vqa_ctx *c)
{
#ifdef VQA_TEST_SEAMS
    if (c->seam_fail_at > 0 && ++c->seam_ensure_calls == c->seam_fail_at) {
        c->last_err = "test seam: device reservation #" + std::to_string(c->seam_fail_at) + " made to fail";
        return true;
    }
#else
    (void)c;
#endif
    return false;
}

static int ensure(vqa_ctx *c, dbuf &b, size_t bytes)
{
    if (seam_reservation_fails(c)) return VQA_ERR_OOM;
    if (bytes <= b.cap) return VQA_OK;
    // contents are scratch; a pending async user is on one of our own streams
    if (int rc = sync_all(c)) return rc;
    if (b.p) HIPCHK(c, hipFree(b.p));
    b.p = nullptr; b.cap = 0;
    const size_t want = bytes + bytes / 8 + 256;
    HIPCHK(c, hipMalloc(&b.p, want));
    b.cap = want;
    return VQA_OK;
}

static int ensure_pinned(vqa_ctx *c, void *&p, size_t &cap, size_t bytes)
{
    if (bytes <= cap) return VQA_OK;
    if (int rc = sync_all(c)) return rc;
    if (p) HIPCHK(c, hipHostFree(p));
    p = nullptr; cap = 0;
    HIPCHK(c, hipHostMalloc(&p, bytes, hipHostMallocDefault));
    cap = bytes;
    return VQA_OK;
}

static void release(dbuf &b)
{
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr; b.cap = 0;
}

// The device arrays of ONE table set while it is being built: whatever was reserved is given back unless the whole set
// made it (commit) - a failure after the first hipMalloc used to drop the earlier pointers.
struct table_builder {
    vqa_ctx *c;
    std::vector<void *> ptrs;
    bool kept = false;
    explicit table_builder(vqa_ctx *c_) : c(c_) {}
    ~table_builder()
    {
        if (!kept)
            for (void *p : ptrs) (void)hipFree(p);
    }
    template <class T> int upload(T **out, const void *src, size_t bytes)
    {
        if (seam_reservation_fails(c)) return VQA_ERR_OOM;
        void *d = nullptr;
        HIPCHK(c, hipMalloc(&d, bytes ? bytes : 1));
        ptrs.push_back(d);
        if (bytes) HIPCHK(c, hipMemcpy(d, src, bytes, hipMemcpyHostToDevice));
        *out = (T *)d;
        return VQA_OK;
    }
    void commit() { kept = true; }
};

static void free_table(resize_tabs &t)
{
    (void)hipFree(t.xofs); (void)hipFree(t.xa); (void)hipFree(t.yofs); (void)hipFree(t.yb);
}
static void free_table(fb_resize_tabs &t)
{
    (void)hipFree(t.xofs); (void)hipFree(t.xa); (void)hipFree(t.yofs); (void)hipFree(t.yb);
    if (t.cols) (void)hipFree(t.cols);
    if (t.rows) (void)hipFree(t.rows);
    if (t.sx) (void)hipFree(t.sx);
    if (t.sy) (void)hipFree(t.sy);
}
static void free_table(float *&m) { (void)hipFree(m); }
static void free_table(dct_fft_plan &P) { (void)hipFree((void *)P.tw); (void)hipFree((void *)P.post); }

// lookup that refreshes the entry's stamp
template <class Map, class Key, class T> static bool cache_get(vqa_ctx *c, Map &m, const Key &key, T &out)
{
    auto it = m.find(key);
    if (it == m.end()) return false;
    it->second.used = ++c->cache_clock;
    out = it->second.v;
    return true;
}

// insert; beyond VQA_TABLE_CACHE_GEOMETRIES entries the least recently used set is freed first.  One submit asks for
// fewer sets of a kind than the bound (<= 6: the Farneback pyramid's), and every set it asked for carries a newer stamp
// than any other, so an eviction can only hit tables of EARLIER submits - whose kernels have completed (a submit is
// refused while another is pending); sync_all before the free is the belt to those braces.
template <class Map, class Key, class T> static int cache_put(vqa_ctx *c, Map &m, const Key &key, const T &v)
{
    while (m.size() >= (size_t)VQA_TABLE_CACHE_GEOMETRIES) {
        auto old = m.begin();
        for (auto it = m.begin(); it != m.end(); ++it)
            if (it->second.used < old->second.used) old = it;
        if (int rc = sync_all(c)) return rc;
        free_table(old->second.v);
        m.erase(old);
    }
    cached<T> e;
    e.v = v;
    e.used = ++c->cache_clock;
    m[key] = e;
    return VQA_OK;
}

// OpenCV resize.cpp coefficient tables for INTER_LINEAR on 8-bit data (see
// DESIGN.md §4.2; restated independently of the oracle's copy).
static void build_axis(int ssize, int dsize, bool is_x, std::vector<int32_t> &ofs, std::vector<int32_t> &coef)
{
    ofs.resize(dsize);
    coef.resize(2 * (size_t)dsize);
    const double scale = 1.0 / ((double)dsize / (double)ssize);
    for (int d = 0; d < dsize; d++) {
        float fr = (float)((d + 0.5) * scale - 0.5);
        int s = (int)std::floor(fr);
        fr -= (float)s;
        if (is_x) {
            if (s < 0) { fr = 0.f; s = 0; }
            if (s >= ssize - 1) { fr = 0.f; s = ssize - 1; }
        }
        ofs[d] = s;
        long a0 = lrintf((1.f - fr) * 2048.f), a1 = lrintf(fr * 2048.f);
        a0 = a0 < -32768 ? -32768 : (a0 > 32767 ? 32767 : a0);
        a1 = a1 < -32768 ? -32768 : (a1 > 32767 ? 32767 : a1);
        coef[2 * (size_t)d] = (int32_t)a0;
        coef[2 * (size_t)d + 1] = (int32_t)a1;
    }
}

static int get_tabs(vqa_ctx *c, int h, int w, int rh, int rw, resize_tabs &out)
{
    auto key = std::make_tuple(h, w, rh, rw);
    if (cache_get(c, c->tabs, key, out)) return VQA_OK;
    resize_tabs t;
    t.mode = (w == 2 * rw && h == 2 * rh) ? 1 : 0;
    std::vector<int32_t> xo, xa, yo, yb;
    build_axis(w, rw, true, xo, xa);
    build_axis(h, rh, false, yo, yb);
    table_builder tb(c);
    int rc;
    if ((rc = tb.upload(&t.xofs, xo.data(), sizeof(int32_t) * rw))) return rc;
    if ((rc = tb.upload(&t.xa, xa.data(), sizeof(int32_t) * 2 * rw))) return rc;
    if ((rc = tb.upload(&t.yofs, yo.data(), sizeof(int32_t) * rh))) return rc;
    if ((rc = tb.upload(&t.yb, yb.data(), sizeof(int32_t) * 2 * rh))) return rc;
    if ((rc = cache_put(c, c->tabs, key, t))) return rc;
    tb.commit();
    out = t;
    return VQA_OK;
}

// orthonormal DCT-II matrix C[k][i] = s_k cos(pi (2i+1) k / 2N)  (what cv2.dct applies)
static int get_dct_matrix(vqa_ctx *c, int n, float **out)
{
    if (cache_get(c, c->dct_mats, n, *out)) return VQA_OK;
    std::vector<float> m((size_t)n * n);
    for (int k = 0; k < n; k++) {
        const double s = k == 0 ? std::sqrt(1.0 / n) : std::sqrt(2.0 / n);
        for (int i = 0; i < n; i++) m[(size_t)k * n + i] = (float)(s * std::cos(M_PI * (2.0 * i + 1.0) * k / (2.0 * n)));
    }
    float *d = nullptr;
    table_builder tb(c);
    int rc;
    if ((rc = tb.upload(&d, m.data(), sizeof(float) * m.size()))) return rc;
    if ((rc = cache_put(c, c->dct_mats, n, d))) return rc;
    tb.commit();
    *out = d;
    return VQA_OK;
}

// k_dct_fft.hip's plan for length n (tables formed in double, rounded once); VQA_ERR_UNSUPPORTED if n does not factor
static int get_fft_plan(vqa_ctx *c, int n, dct_fft_plan *out)
{
    if (cache_get(c, c->fft_plans, n, *out)) return VQA_OK;
    dct_fft_plan P;
    memset(&P, 0, sizeof P);
    P.n = n;
    if (!dct_fft_factor(n, P.radix, &P.npass)) return VQA_ERR_UNSUPPORTED;
    for (int p = 0, Ns = 1; p < P.npass; Ns *= P.radix[p], p++) {
        P.m[p] = n / P.radix[p];
        P.tstep[p] = n / (Ns * P.radix[p]);
        // j / Ns = mulhi(j, ceil(2^32 / Ns)), exact while j * Ns < 2^32 (n <= 4000); the first pass (Ns = 1) does not use it
        P.ns_magic[p] = Ns == 1 ? 0u : (uint32_t)((0x100000000ull + (uint64_t)Ns - 1) / (uint64_t)Ns);
    }
    std::vector<float2> tw((size_t)n), post((size_t)n);
    for (int m = 0; m < n; m++) {
        const double a = -2.0 * M_PI * (double)m / (double)n;
        tw[m] = make_float2((float)std::cos(a), (float)std::sin(a));
        const double s = m == 0 ? std::sqrt(1.0 / n) : std::sqrt(2.0 / n), b = M_PI * (double)m / (2.0 * n);
        post[m] = make_float2((float)(s * std::cos(b)), (float)(s * std::sin(b)));
    }
    float2 *dtw = nullptr, *dpost = nullptr;
    table_builder tb(c);
    int rc;
    if ((rc = tb.upload(&dtw, tw.data(), sizeof(float2) * n))) return rc;
    if ((rc = tb.upload(&dpost, post.data(), sizeof(float2) * n))) return rc;
    P.tw = dtw; P.post = dpost;
    if ((rc = cache_put(c, c->fft_plans, n, P))) return rc;
    tb.commit();
    *out = P;
    return VQA_OK;
}

// ---- Farneback (k_farneback.hip): host-side constants, tables and the level loop ---------------------
// getGaussianKernel(ksize, sigma, CV_32F): taps formed and normalised in double, rounded to float; the
// fixed 1/4,1/2,1/4 kernel when sigma == 0 (restated independently of the oracle's copy)
static fb_taps fb_gauss_taps(int ksize, double sigma)
{
    fb_taps T;
    memset(&T, 0, sizeof T);
    T.ksize = ksize;
    if (sigma <= 0 && ksize == 3) { T.k[0] = 0.25f; T.k[1] = 0.5f; T.k[2] = 0.25f; return T; }
    double kd[32], sum = 0;
    const int r = ksize / 2;
    const double sc = -0.5 / (sigma * sigma);
    for (int i = 0; i < ksize; i++) { const double x = i - r; kd[i] = std::exp(sc * x * x); sum += kd[i]; }
    for (int i = 0; i < ksize; i++) T.k[i] = (float)(kd[i] / sum);
    return T;
}

// FarnebackPrepareGaussian(5, 1.2): taps and the four needed entries of inv(G) (G is block-sparse: closed form)
static fb_poly fb_poly_consts()
{
    fb_poly C;
    const int n = 5;
    const double sigma = 1.2;
    double s = 0;
    for (int x = -n; x <= n; x++) { C.g[x + n] = (float)std::exp(-x * x / (2 * sigma * sigma)); s += C.g[x + n]; }
    s = 1. / s;
    for (int x = -n; x <= n; x++) {
        C.g[x + n] = (float)(C.g[x + n] * s);
        C.xg[x + n] = (float)(x * C.g[x + n]);
        C.xxg[x + n] = (float)(x * x * C.g[x + n]);
    }
    double a = 0, b = 0, c = 0, d = 0;
    for (int y = -n; y <= n; y++)
        for (int x = -n; x <= n; x++) {
            const double gg = (double)C.g[y + n] * C.g[x + n];
            a += gg; b += gg * x * x; c += gg * x * x * x * x; d += gg * x * x * y * y;
        }
    const double det = a * (c * c - d * d) - 2 * b * b * (c - d);
    C.ig11 = 1. / b;
    C.ig03 = -b * (c - d) / det;
    C.ig33 = (a * c - b * b) / det;
    C.ig55 = 1. / d;
    return C;
}

static void fb_build_axis(int ssize, int dsize, bool is_x, std::vector<int32_t> &ofs, std::vector<float> &coef)
{
    ofs.resize(dsize);
    coef.resize(2 * (size_t)dsize);
    const double scale = 1.0 / ((double)dsize / (double)ssize);
    for (int d = 0; d < dsize; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)std::floor(f);
        f -= (float)s;
        if (is_x) {
            if (s < 0) { f = 0.f; s = 0; }
            if (s >= ssize - 1) { f = 0.f; s = ssize - 1; }
        }
        ofs[d] = s;
        coef[2 * (size_t)d] = 1.f - f;
        coef[2 * (size_t)d + 1] = f;
    }
}

static int get_fb_tabs(vqa_ctx *c, int sh, int sw, int dh, int dw, fb_resize_tabs &out)
{
    auto key = std::make_tuple(sh, sw, dh, dw);
    if (cache_get(c, c->fb_tabs, key, out)) return VQA_OK;
    fb_resize_tabs t;
    t.mode = (sw == 2 * dw && sh == 2 * dh) ? 1 : 0;
    std::vector<int32_t> xo, yo;
    std::vector<float> xa, yb;
    fb_build_axis(sw, dw, true, xo, xa);
    fb_build_axis(sh, dh, false, yo, yb);
    table_builder tb(c);
    int rc;
    if ((rc = tb.upload(&t.xofs, xo.data(), sizeof(int32_t) * dw))) return rc;
    if ((rc = tb.upload(&t.xa, xa.data(), sizeof(float) * 2 * dw))) return rc;
    if ((rc = tb.upload(&t.yofs, yo.data(), sizeof(int32_t) * dh))) return rc;
    if ((rc = tb.upload(&t.yb, yb.data(), sizeof(float) * 2 * dh))) return rc;
    if (dw <= sw && dh <= sh) {
        // the samples of each level column / row as the fused level kernel wants them, and their extent per tile
        std::vector<int32_t> sx(2 * (size_t)dw), sy(2 * (size_t)dh);
        for (int d = 0; d < dw; d++) {
            sx[2 * d] = t.mode == 1 ? 2 * d : xo[d];
            sx[2 * d + 1] = t.mode == 1 ? 2 * d + 1 : (xo[d] + 1 < sw ? xo[d] + 1 : sw - 1);
        }
        for (int d = 0; d < dh; d++) {
            const int y0 = yo[d] < 0 ? 0 : (yo[d] > sh - 1 ? sh - 1 : yo[d]);
            const int y1 = yo[d] + 1 < 0 ? 0 : (yo[d] + 1 > sh - 1 ? sh - 1 : yo[d] + 1);
            sy[2 * d] = t.mode == 1 ? 2 * d : y0;
            sy[2 * d + 1] = t.mode == 1 ? 2 * d + 1 : y1;
        }
        auto span = [](const std::vector<int32_t> &v, int n, int tile) {
            int m = 0;
            bool mono = true;
            for (size_t i = 1; i < v.size(); i++) mono = mono && v[i] >= v[i - 1];
            if (!mono) return -1;
            for (int a = 0; a < n; a++) { // every window of `tile` consecutive level columns / rows, aligned or not
                const int b = (a + tile < n ? a + tile : n) - 1;
                const int e = v[2 * b + 1] - v[2 * a] + 1;
                m = e > m ? e : m;
            }
            return m;
        };
        t.span_x32 = span(sx, dw, 32); t.span_y8 = span(sy, dh, 8); t.span_y32 = span(sy, dh, 32);
        if (t.span_x32 > 0 && t.span_y8 > 0 && t.span_y32 > 0) {
            if ((rc = tb.upload(&t.sx, sx.data(), sizeof(int32_t) * sx.size()))) return rc;
            if ((rc = tb.upload(&t.sy, sy.data(), sizeof(int32_t) * sy.size()))) return rc;
        }
    }
    if (t.mode == 0 && (dw < sw || dh < sh)) {
        // what the bilinear taps read: columns xofs, min(xofs + 1, sw - 1); rows clamp(yofs), clamp(yofs + 1)
        std::vector<int32_t> cs, rs;
        std::vector<char> cm(sw, 0), rm(sh, 0);
        for (int d = 0; d < dw; d++) { cm[xo[d]] = 1; cm[xo[d] + 1 < sw ? xo[d] + 1 : sw - 1] = 1; }
        for (int d = 0; d < dh; d++) {
            const int y0 = yo[d] < 0 ? 0 : (yo[d] > sh - 1 ? sh - 1 : yo[d]);
            const int y1 = yo[d] + 1 < 0 ? 0 : (yo[d] + 1 > sh - 1 ? sh - 1 : yo[d] + 1);
            rm[y0] = 1; rm[y1] = 1;
        }
        for (int x = 0; x < sw; x++) if (cm[x]) cs.push_back(x);
        for (int y = 0; y < sh; y++) if (rm[y]) rs.push_back(y);
        t.nc = (int)cs.size(); t.nr = (int)rs.size();
        if ((rc = tb.upload(&t.cols, cs.data(), sizeof(int32_t) * t.nc))) return rc;
        if ((rc = tb.upload(&t.rows, rs.data(), sizeof(int32_t) * t.nr))) return rc;
    }
    if ((rc = cache_put(c, c->fb_tabs, key, t))) return rc;
    tb.commit();
    out = t;
    return VQA_OK;
}

// gray: n + 1 planes (slot 0 = the frame before the batch); pair i = planes i, i + 1 -> res[i].flow_mag_mean.
// Pairs are processed in chunks whose planes (expansions: 20 B/pixel) stay resident in a bounded scratch.
static int run_farneback(vqa_ctx *c, hipStream_t st, const uint8_t *gray, int gp, int64_t plane_stride, int n, int h,
                         int w, bool first_has_prev, vqa_frame_metrics *res)
{
    const double pyr_scale = 0.5;
    const int iters = 3, min_size = 32;
    int levels = 3, k;
    double scale = 1;
    for (k = 0; k < levels; k++) {
        scale *= pyr_scale;
        if (w * scale < min_size || h * scale < min_size) break;
    }
    levels = k;
    const fb_poly PC = fb_poly_consts();
    const size_t P = (size_t)h * w;
    // per pair ~59 B/pixel of scratch (planes: blur tmp 4 + blurred 4 + level image 4 + expansions of every level 26.7;
    // pairs: two flow fields 16; + 20 for the products of the lab build's two-kernel form); a chunk stays under ~12 GiB of the
    // 288: a 64-frame 1080p batch is ONE chunk (round 3's 3 GiB cut it into 21 + 21 + 21 + 1 pairs, and the
    // one-pair tail ran fifty launches on an empty chip)
#ifdef VQA_AB_VARIANTS
    const bool two_kernel = ab_knob("VQA_FB_VARIANT", 0) == 1;
#else
    const bool two_kernel = false;
#endif
    // the level image: one fused kernel (blur + resize); the three-kernel form is the fallback for levels whose patch
    // exceeds the LDS budget (and VQA_FB_LEVEL_VARIANT=1 in the lab build)
#ifdef VQA_AB_VARIANTS
    const bool three_kernel = ab_knob("VQA_FB_LEVEL_VARIANT", 0) == 1;
#else
    const bool three_kernel = false;
#endif
    // Level geometry (coarse to fine: k = levels .. 0) and where each level's expansions live: with VQA_OPT_OVERLAP the level
    // images and expansions of ALL levels are formed on a side stream (they depend on the gray planes only) while the main
    // stream runs the flow iterations of the coarser levels - the finest level's expansion, a quarter of the pre-pass work,
    // is ready by the time the iterations reach it.  Each level keeps its own expansion planes for that (4/3 of one level's).
    int lwk[4], lhk[4], ksz[4];
    double sig[4];
    size_t Roff[4], Rtot = 0;
    for (k = levels; k >= 0; k--) {
        scale = 1;
        for (int i = 0; i < k; i++) scale *= pyr_scale;
        sig[k] = (1. / scale - 1) * 0.5;
        ksz[k] = (int)std::lrint(sig[k] * 5) | 1;
        if (ksz[k] < 3) ksz[k] = 3;
        lwk[k] = (int)std::lrint(w * scale);
        lhk[k] = (int)std::lrint(h * scale);
    }
    // pairs per chunk: ~12 GiB of scratch, and never more than 80 % of what the device can give right now (what this ctx
    // already holds for Farneback counts as available: ensure() re-uses or replaces it).  If a reservation still fails -
    // another context or process took the memory in between - the chunk is halved and everything re-reserved, down to
    // one pair, before the submit reports VQA_ERR_OOM.
    dbuf *const fb_bufs[] = {&c->fb_tmp, &c->fb_blur, &c->fb_img, &c->fb_R, &c->fb_M, &c->fb_flow0, &c->fb_flow1, &c->fb_part};
    const size_t per_pair = (size_t)(two_kernel ? 72 : 59) * P;
    size_t budget = 12ull << 30;
    if (c->seam_fb_chunk_bytes > 0) budget = (size_t)c->seam_fb_chunk_bytes; // (lab build only: never set in the shipped library)
    {
        size_t free_b = 0, total_b = 0, held = 0;
        for (dbuf *b : fb_bufs) held += b->cap;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            const size_t avail = (size_t)((double)(free_b + held) * 0.8);
            if (avail < budget) budget = avail;
        } else {
            (void)hipGetLastError();
        }
    }
    int mc = (int)(budget / per_pair);
    mc = mc < 1 ? 1 : (mc > n ? n : mc);
    const bool piped = c->opt_overlap && !two_kernel && !three_kernel;
    int rc;
    for (;;) {
        Rtot = 0;
        for (k = levels; k >= 0; k--) { Roff[k] = Rtot; Rtot += (size_t)5 * lhk[k] * lwk[k] * (mc + 1); }
        int nb = fb_iter_max_blocks(h, w); // partial |flow| sums per pair: one per workgroup of the last iteration
#ifdef VQA_AB_VARIANTS
        if (fb_mag_blocks() > nb) nb = fb_mag_blocks();
#endif
        rc = ensure(c, c->fb_tmp, sizeof(float) * P * (mc + 1));
        if (!rc) rc = ensure(c, c->fb_blur, sizeof(float) * P * (mc + 1));
        if (!rc) rc = ensure(c, c->fb_img, sizeof(float) * P * (mc + 1));
        if (!rc) rc = ensure(c, c->fb_R, sizeof(float) * Rtot);
        if (!rc && two_kernel) rc = ensure(c, c->fb_M, sizeof(float) * 5 * P * mc);
        if (!rc) rc = ensure(c, c->fb_flow0, sizeof(float) * 2 * P * mc);
        if (!rc) rc = ensure(c, c->fb_flow1, sizeof(float) * 2 * P * mc);
        if (!rc) rc = ensure(c, c->fb_part, sizeof(double) * (size_t)nb * mc);
        if (rc != VQA_ERR_OOM || mc == 1 || c->last_err.compare(0, 10, "test seam:") == 0) break; // (an injected failure is not retried)
        (void)hipGetLastError();
        if (int rs = sync_all(c)) return rs;
        for (dbuf *b : fb_bufs) release(*b);
        mc = (mc + 1) / 2;
    }
    if (rc) return rc;
    float *tmp = (float *)c->fb_tmp.p, *blur = (float *)c->fb_blur.p, *img = (float *)c->fb_img.p;
    float *Rall = (float *)c->fb_R.p;
#ifdef VQA_AB_VARIANTS
    float *M = (float *)c->fb_M.p;
#endif
    float *flow = (float *)c->fb_flow0.p, *prev_flow = (float *)c->fb_flow1.p;
    hipStream_t sp = st; // the stream of the level images and expansions
    if (piped) {
        if (!c->side[0]) HIPCHK(c, hipStreamCreateWithFlags(&c->side[0], hipStreamNonBlocking)); // block-SAD's stream: idle in this mode
        for (int i = 0; i < 6; i++)
            if (!c->fb_ev[i]) HIPCHK(c, hipEventCreateWithFlags(&c->fb_ev[i], hipEventDisableTiming));
        sp = c->side[0];
        HIPCHK(c, hipEventRecord(c->fb_ev[4], st)); // the gray planes exist in st's order
        HIPCHK(c, hipStreamWaitEvent(sp, c->fb_ev[4], 0));
    }
    // the level image and the expansion of level k for the chunk's planes, on stream s
    auto expand = [&](hipStream_t s, int k, const uint8_t *g0, int planes) -> int {
        const int lw = lwk[k], lh = lhk[k];
        const float *level_img = img;
        const fb_taps K = fb_gauss_taps(ksz[k], sig[k]);
        if (lw != w || lh != h) {
            fb_resize_tabs T;
            if (int r2 = get_fb_tabs(c, h, w, lh, lw, T)) return r2;
            if (three_kernel || !launch_fb_level(s, g0, gp, plane_stride, planes, h, w, K, &T, img, lh, lw)) {
                launch_fb_blur(s, g0, gp, plane_stride, planes, h, w, K, T.cols, T.nc, T.rows, T.nr, tmp, blur);
                launch_fb_resize(s, blur, h, w, 1, img, lh, lw, planes, T, 1.f, false);
            }
        } else if (three_kernel || !launch_fb_level(s, g0, gp, plane_stride, planes, h, w, K, nullptr, img, lh, lw)) {
            launch_fb_blur(s, g0, gp, plane_stride, planes, h, w, K, nullptr, 0, nullptr, 0, tmp, blur);
            level_img = blur;
        }
        launch_fb_polyexp(s, level_img, planes, lh, lw, PC, Rall + Roff[k]);
        return VQA_OK;
    };
    for (int a = 0; a < n; a += mc) {
        const int pairs = (n - a) < mc ? (n - a) : mc, planes = pairs + 1;
        const uint8_t *g0 = gray + (int64_t)a * plane_stride;
        if (piped) {
            // the previous chunk's iterations still read the expansion planes: the side stream waits for them
            if (a > 0) HIPCHK(c, hipStreamWaitEvent(sp, c->fb_ev[5], 0));
            for (k = levels; k >= 0; k--) {
                if ((rc = expand(sp, k, g0, planes))) return rc;
                HIPCHK(c, hipEventRecord(c->fb_ev[k], sp));
            }
        }
        int pw = 0, ph = 0;
        for (k = levels; k >= 0; k--) {
            const int lw = lwk[k], lh = lhk[k];
            const float *R = Rall + Roff[k];
            // flow of this level: zero at the coarsest, else the coarser level's flow resized and doubled
            fb_resize_tabs TF;
            const bool coarsest = (k == levels);
            // exact 2x decimation cannot occur here (this is an upscale), so the bilinear tables always apply
            if (!coarsest && (rc = get_fb_tabs(c, ph, pw, lh, lw, TF))) return rc;
            // every plane once: blur (only where the resize will sample), resize to the level, polynomial expansion
            if (piped) HIPCHK(c, hipStreamWaitEvent(st, c->fb_ev[k], 0));
            else if ((rc = expand(st, k, g0, planes))) return rc;
#ifdef VQA_AB_VARIANTS
            if (two_kernel) {
                launch_fb_update_first(st, R, coarsest ? nullptr : prev_flow, ph, pw, TF, (float)(1. / pyr_scale), pairs, lh, lw, M);
                for (int i = 0; i < iters; i++) {
                    launch_fb_blur_solve(st, M, pairs, lh, lw, flow);
                    if (i < iters - 1) launch_fb_update(st, R, flow, pairs, lh, lw, M);
                }
                float *t = flow; flow = prev_flow; prev_flow = t; // prev_flow = this level's result
            } else
#endif
            {
                // The coarser level's result (prev_flow) is upsampled and doubled into `flow` by the upsample kernel (8 B/pixel
                // written once per level: forming it inside the first iteration instead cost that iteration 70 % more time
                // than the write - the eight dependent loads per row sit on the march's critical path); the iterations then
                // ping-pong between the two buffers, and prev_flow ends up naming this level's result.
                const float *in = nullptr;
                if (!coarsest) {
                    launch_fb_resize(st, prev_flow, ph, pw, 2, flow, lh, lw, pairs, TF, (float)(1. / pyr_scale), true);
                    float *t = flow; flow = prev_flow; prev_flow = t;
                    in = prev_flow;
                }
                for (int i = 0; i < iters; i++) {
                    // the last iteration of the finest level also leaves the partial sums of |flow| (no magnitude pass)
                    const bool last = k == 0 && i == iters - 1;
                    launch_fb_iter(st, R, in, pairs, lh, lw, flow, last ? (double *)c->fb_part.p : nullptr);
                    float *t = flow; flow = prev_flow; prev_flow = t;
                    in = prev_flow;
                }
            }
            pw = lw; ph = lh;
        }
#ifdef VQA_AB_VARIANTS
        if (two_kernel) launch_fb_mag(st, prev_flow, pairs, h, w, (double *)c->fb_part.p, a > 0 || first_has_prev, res + a);
        else
#endif
        launch_fb_mag_finalize(st, (const double *)c->fb_part.p, fb_iter_blocks(pairs, h, w), pairs, h, w, a > 0 || first_has_prev,
                               res + a);
        if (piped && a + mc < n) HIPCHK(c, hipEventRecord(c->fb_ev[5], st));
    }
    return VQA_OK;
}

extern "C" {

int vqa_abi_version(void) { return VQA_ABI_VERSION; }

const char *vqa_strerror(int s)
{
    switch (s) {
    case VQA_OK: return "ok";
    case VQA_ERR_INVALID: return "invalid argument";
    case VQA_ERR_NO_DEVICE: return "no HIP device (this engine has no CPU fallback)";
    case VQA_ERR_HIP: return "HIP runtime error";
    case VQA_ERR_OOM: return "out of memory";
    case VQA_ERR_UNSUPPORTED: return "unsupported request";
    case VQA_ERR_STATE: return "call sequence error";
    case VQA_ERR_INCOMPLETE: return "a result would be inexact and is withheld";
    default: return "unknown status";
    }
}

int vqa_device_count(int *count)
{
    if (!count) return VQA_ERR_INVALID;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { *count = 0; return VQA_ERR_NO_DEVICE; }
    *count = n;
    return VQA_OK;
}

void vqa_default_params(vqa_params *p)
{
    if (!p) return;
    memset(p, 0, sizeof *p);
    p->canny_low = 100;
    p->canny_high = 200;
    p->sad_range = 7;
    p->dct_mode = VQA_DCT_AUTO;
}

int vqa_create(int device, vqa_ctx **out)
{
    if (!out) return VQA_ERR_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return VQA_ERR_NO_DEVICE;
    if (device < 0 || device >= n) return VQA_ERR_NO_DEVICE;
    vqa_ctx *c = new (std::nothrow) vqa_ctx();
    if (!c) return VQA_ERR_OOM;
    c->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return VQA_ERR_HIP;
    }
    c->dct_wave_slots = dct8_wave_slots(); // (the device is current: hipSetDevice above)
    // the one environment variable the shipped library reads (documented in vqa.h): the initial VQA_OPT_OVERLAP
    if (const char *e = getenv("VQA_OVERLAP")) c->opt_overlap = atoi(e) != 0;
#ifdef VQA_TEST_SEAMS
    if (const char *e = getenv("VQA_HYST_MAX_ROUNDS")) c->seam_hyst_max_rounds = atoi(e) > 0 ? atoi(e) : 0;
    if (const char *e = getenv("VQA_HYST_RESCUE_MAX_ROUNDS")) c->seam_hyst_rescue_max_rounds = atoi(e) > 0 ? atoi(e) : 0;
    if (const char *e = getenv("VQA_FAIL_ENSURE_AT")) c->seam_fail_at = atol(e);
    if (const char *e = getenv("VQA_FB_CHUNK_BYTES")) c->seam_fb_chunk_bytes = atoll(e);
#endif
    *out = c;
    return VQA_OK;
}

int vqa_set_option(vqa_ctx *c, int option, int value)
{
    if (!c) return VQA_ERR_INVALID;
    if (c->pend_c || c->pend_q) return VQA_ERR_STATE; // options apply to whole submits
    switch (option) {
    case VQA_OPT_OVERLAP: c->opt_overlap = value != 0; return VQA_OK;
    case VQA_OPT_HYST_STATS: c->opt_hyst_stats = value != 0; return VQA_OK;
    default: return VQA_ERR_INVALID;
    }
}

int vqa_get_option(const vqa_ctx *c, int option, int *value)
{
    if (!c || !value) return VQA_ERR_INVALID;
    switch (option) {
    case VQA_OPT_OVERLAP: *value = c->opt_overlap ? 1 : 0; return VQA_OK;
    case VQA_OPT_HYST_STATS: *value = c->opt_hyst_stats ? 1 : 0; return VQA_OK;
    default: return VQA_ERR_INVALID;
    }
}

int vqa_build_flavour(void)
{
    int f = 0;
#ifdef VQA_AB_VARIANTS
    f |= 1;
#endif
#ifdef VQA_TEST_SEAMS
    f |= 2;
#endif
    return f;
}

// every grow-only buffer, the result staging and every cached table of an idle ctx (vqa_trim, vqa_destroy)
static void release_scratch(vqa_ctx *c)
{
    dbuf *bufs[] = {&c->gray_full, &c->planeA, &c->planeB, &c->state, &c->res_dev, &c->partials, &c->tile_flags,
                    &c->dirty0, &c->dirty1, &c->again_dev, &c->stage_frames, &c->stage_prev, &c->dct_scratch,
                    &c->dct_pe, &c->dct_pt, &c->qres_dev, &c->qpartials, &c->qstage_ref, &c->qstage_dist,
                    &c->fb_tmp, &c->fb_blur, &c->fb_img, &c->fb_R, &c->fb_M, &c->fb_flow0, &c->fb_flow1, &c->fb_part};
    for (dbuf *b : bufs) release(*b);
    for (auto &kv : c->tabs) free_table(kv.second.v);
    for (auto &kv : c->fb_tabs) free_table(kv.second.v);
    for (auto &kv : c->dct_mats) free_table(kv.second.v);
    for (auto &kv : c->fft_plans) free_table(kv.second.v);
    c->tabs.clear(); c->fb_tabs.clear(); c->dct_mats.clear(); c->fft_plans.clear();
    if (c->res_host) (void)hipHostFree(c->res_host);
    if (c->qres_host) (void)hipHostFree(c->qres_host);
    c->res_host = c->qres_host = nullptr;
    c->res_host_cap = c->qres_host_cap = 0;
    // the planes vqa_debug_read_plane would read are gone
    c->last_n = 0; c->last_has_full = c->last_has_state = c->last_has_planes = false;
}

int vqa_trim(vqa_ctx *c)
{
    if (!c) return VQA_ERR_INVALID;
    if (c->pend_c || c->pend_q) return VQA_ERR_STATE;
    HIPCHK(c, hipSetDevice(c->device));
    if (int rc = sync_all(c)) return rc;
    prof_collect(c);
    release_scratch(c);
    (void)hipGetLastError();
    return VQA_OK;
}

int vqa_destroy(vqa_ctx *c)
{
    if (!c) return VQA_ERR_INVALID;
    (void)hipSetDevice(c->device);
    (void)sync_all(c);
    release_scratch(c);
    for (auto &t : c->ev_open) { (void)hipEventDestroy(std::get<1>(t)); (void)hipEventDestroy(std::get<2>(t)); }
    for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
    for (int i = 0; i < 3; i++) {
        if (c->side[i]) (void)hipStreamDestroy(c->side[i]);
        if (c->ev_join[i]) (void)hipEventDestroy(c->ev_join[i]);
    }
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_xctx) (void)hipEventDestroy(c->ev_xctx);
    for (int i = 0; i < 6; i++)
        if (c->fb_ev[i]) (void)hipEventDestroy(c->fb_ev[i]);
    (void)hipStreamDestroy(c->stream);
    delete c;
    return VQA_OK;
}

const char *vqa_last_hip_error(const vqa_ctx *c) { return c ? c->last_err.c_str() : ""; }

int vqa_alloc_pinned(vqa_ctx *c, size_t bytes, void **out)
{
    if (!c || !out || !bytes) return VQA_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipHostMalloc(out, bytes, hipHostMallocDefault));
    return VQA_OK;
}
int vqa_free_pinned(vqa_ctx *c, void *p)
{
    if (!c || !p) return VQA_ERR_INVALID;
    HIPCHK(c, hipHostFree(p));
    return VQA_OK;
}
static bool host_byte_is_pinned(const void *p)
{
    hipPointerAttribute_t a;
    memset(&a, 0, sizeof a);
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { // ordinary host memory: HIP has never heard of it
        (void)hipGetLastError();
        return false;
    }
    return a.type == hipMemoryTypeHost;
}
int vqa_host_is_pinned(vqa_ctx *c, const void *p, size_t bytes, int *out)
{
    if (!c || !p || !out || !bytes) return VQA_ERR_INVALID;
    // the first AND the last byte: an array that starts inside a registered region and ends outside it (hipHostRegister
    // of part of a buffer) is pageable as far as a DMA of the whole range is concerned
    *out = (host_byte_is_pinned(p) && host_byte_is_pinned((const uint8_t *)p + (bytes - 1))) ? 1 : 0;
    return VQA_OK;
}
int vqa_alloc_device(vqa_ctx *c, size_t bytes, void **out)
{
    if (!c || !out || !bytes) return VQA_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMalloc(out, bytes));
    return VQA_OK;
}
int vqa_free_device(vqa_ctx *c, void *p)
{
    if (!c || !p) return VQA_ERR_INVALID;
    HIPCHK(c, hipFree(p));
    return VQA_OK;
}
int vqa_copy_h2d(vqa_ctx *c, void *dst, const void *src, size_t bytes)
{
    if (!c || !dst || !src) return VQA_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    return VQA_OK;
}
int vqa_stream_wait(vqa_ctx *waiter, vqa_ctx *signaler)
{
    if (!waiter || !signaler || waiter == signaler) return VQA_ERR_INVALID;
    if (waiter->device != signaler->device) return VQA_ERR_INVALID;
    HIPCHK(waiter, hipSetDevice(waiter->device));
    if (!signaler->ev_xctx) HIPCHK(waiter, hipEventCreateWithFlags(&signaler->ev_xctx, hipEventDisableTiming));
    // the record captures the signaler's stream as it stands NOW; re-recording the event later does not move a wait already enqueued
    HIPCHK(waiter, hipEventRecord(signaler->ev_xctx, signaler->stream));
    HIPCHK(waiter, hipStreamWaitEvent(waiter->stream, signaler->ev_xctx, 0));
    return VQA_OK;
}
int vqa_copy_d2h(vqa_ctx *c, void *dst, const void *src, size_t bytes)
{
    if (!c || !dst || !src) return VQA_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    return VQA_OK;
}
int vqa_sync(vqa_ctx *c)
{
    if (!c) return VQA_ERR_INVALID;
    return sync_all(c);
}
void *vqa_stream(vqa_ctx *c) { return c ? (void *)c->stream : nullptr; }
// internal accessors for vqa_comm.hip (not part of the ABI: hidden visibility)
extern "C" __attribute__((visibility("hidden"))) int vqa_ctx_device_(const vqa_ctx *c) { return c->device; }
extern "C" __attribute__((visibility("hidden"))) void *vqa_ctx_stream_(const vqa_ctx *c) { return (void *)c->stream; }

// ---------------------------------------------------------------------------
// A submit that fails after it has started to enqueue must not hand the caller's buffers (and this ctx's scratch) back
// while kernels or copies still use them: whatever the failing step was - a reservation, a copy, a launch, in the first
// slice or a later one, before or after the fork onto the side streams - the public entry points below drain EVERY
// stream of the ctx, drop the submit's timing events and leave the ctx idle and usable (pend_* stay 0).  The reference's
// convention for a failed step is log-and-re-raise with nothing left running (video_processing.py:295-297).
static int drain_failed_submit(vqa_ctx *c, int rc, bool touched)
{
    if (rc != VQA_OK && touched) {
        const std::string why = c->last_err; // keep the first error's text
        (void)sync_all(c);
        (void)hipGetLastError();
        prof_discard(c);
        c->last_err = why;
    }
    return rc;
}

static int complexity_submit_body(vqa_ctx *c, const uint8_t *frames, const uint8_t *prev0, int mem_kind, int n, int h, int w,
                                  int64_t frame_stride, int64_t row_stride, uint32_t mask, const vqa_params *params,
                                  bool &touched)
{
    if (!c || !frames || n <= 0 || h <= 0 || w <= 0) return VQA_ERR_INVALID;
    if (mem_kind != VQA_MEM_HOST && mem_kind != VQA_MEM_DEVICE) return VQA_ERR_INVALID;
    if (!mask || (mask & ~VQA_M_ALL)) return VQA_ERR_INVALID;
    if (row_stride < (int64_t)3 * w || frame_stride < row_stride * (h - 1) + (int64_t)3 * w) return VQA_ERR_INVALID;
    if (c->pend_c) return VQA_ERR_STATE;
    vqa_params P;
    if (params) P = *params; else vqa_default_params(&P);
    for (int i = 0; i < 9; i++)
        if (P.reserved[i]) return VQA_ERR_INVALID;
    if (P.motion_mode != VQA_MOTION_SAD && P.motion_mode != VQA_MOTION_FARNEBACK) return VQA_ERR_INVALID;
    if (P.sad_range < 0 || P.sad_range > 7) return VQA_ERR_INVALID;
    if (P.resize_w < 0 || P.resize_h < 0 || ((P.resize_w == 0) != (P.resize_h == 0))) return VQA_ERR_INVALID;
    if (P.dct_mode < VQA_DCT_AUTO || P.dct_mode > VQA_DCT_FULL) return VQA_ERR_INVALID;
    if ((int64_t)h * w > (1ll << 28)) return VQA_ERR_UNSUPPORTED;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    touched = true; // from here on a failure is drained by the caller (drain_failed_submit)

    const int rw = P.resize_w ? P.resize_w : w, rh = P.resize_h ? P.resize_h : h;
    const bool resized = !(rw == w && rh == h);
    const int ph = rh, pw = rw;
    const int pp = align_up(pw, 64), gp = align_up(w, 64);
    const bool want_gh = mask & VQA_M_GRAY_HIST, want_ch = mask & VQA_M_COLOR_HIST, want_dct = mask & VQA_M_DCT;
    const bool want_t = mask & VQA_M_TEMPORAL_DCT, want_e = mask & VQA_M_EDGE, want_m = mask & VQA_M_MOTION;
    const bool want_orb = mask & VQA_M_ORB;
    const bool batch_has_prev0 = prev0 != nullptr;

    // ---- bring frames to the device if they are on the host
    const uint8_t *dframes = frames, *dprev = prev0;
    if (mem_kind == VQA_MEM_HOST) {
        // The device copy is always compact (rows of 3w bytes): padded rows / a region of interest inside
        // larger frames cross PCIe as a 2-D copy, so no byte outside the caller's rows is ever read.
        const size_t rbytes = (size_t)3 * w, fbytes = rbytes * h;
        const bool padded = (size_t)row_stride != rbytes;
        int rc = ensure(c, c->stage_frames, fbytes * n);
        if (rc) return rc;
        if (!padded && (size_t)frame_stride == fbytes) {
            HIPCHK(c, hipMemcpyAsync(c->stage_frames.p, frames, fbytes * n, hipMemcpyHostToDevice, st));
        } else {
            // strided selection (every k-th frame of a clip): only the selected frames cross PCIe
            for (int i = 0; i < n; i++) {
                uint8_t *dst = (uint8_t *)c->stage_frames.p + fbytes * i;
                const uint8_t *src = frames + (int64_t)i * frame_stride;
                if (padded)
                    HIPCHK(c, hipMemcpy2DAsync(dst, rbytes, src, (size_t)row_stride, rbytes, h, hipMemcpyHostToDevice, st));
                else
                    HIPCHK(c, hipMemcpyAsync(dst, src, fbytes, hipMemcpyHostToDevice, st));
            }
        }
        dframes = (const uint8_t *)c->stage_frames.p;
        if (batch_has_prev0) {
            rc = ensure(c, c->stage_prev, fbytes);
            if (rc) return rc;
            if (padded)
                HIPCHK(c, hipMemcpy2DAsync(c->stage_prev.p, rbytes, prev0, (size_t)row_stride, rbytes, h,
                                           hipMemcpyHostToDevice, st));
            else
                HIPCHK(c, hipMemcpyAsync(c->stage_prev.p, prev0, fbytes, hipMemcpyHostToDevice, st));
            dprev = (const uint8_t *)c->stage_prev.p;
        }
        frame_stride = (int64_t)fbytes;
        row_stride = (int64_t)rbytes;
    }

    int rc = ensure(c, c->res_dev, sizeof(vqa_frame_metrics) * (size_t)n);
    if (rc) return rc;
    rc = ensure_pinned(c, c->res_host, c->res_host_cap, sizeof(vqa_frame_metrics) * (size_t)n);
    if (rc) return rc;
    HIPCHK(c, hipMemsetAsync(c->res_dev.p, 0, sizeof(vqa_frame_metrics) * (size_t)n, st));
    vqa_frame_metrics *const res_all = (vqa_frame_metrics *)c->res_dev.p;
    const uint8_t *const dframes_all = dframes, *const dprev_all = dprev;
    const int n_all = n;

    // Frames ride in gridDim.y (<= 65535): larger batches are enqueued as consecutive slices on the same stream.  A
    // slice's "previous frame" is the last frame of the slice before it; scratch planes are reused (stream order).
    const int SLICE = 32768;
    auto run_slice = [&](const int a0) -> int {
    const int n = n_all - a0 < SLICE ? n_all - a0 : SLICE;
    const uint8_t *dframes = dframes_all + (int64_t)a0 * frame_stride;
    const uint8_t *dprev = a0 ? dframes_all + (int64_t)(a0 - 1) * frame_stride : dprev_all;
    const bool has_prev0 = a0 > 0 || batch_has_prev0;
    vqa_frame_metrics *res = res_all + a0;

    // ---- planes
    const bool want_plane = want_gh || want_ch || want_dct || want_t || want_e; // kernels that read the configured plane
    const bool need_full = (!resized && want_plane) || want_m;
    const bool need_planes = resized && want_plane;
    const int64_t full_stride = (int64_t)h * gp, plane_stride = (int64_t)ph * pp;
    uint8_t *gfull = nullptr, *pA = nullptr, *pB = nullptr;
    if (need_full) {
        rc = ensure(c, c->gray_full, (size_t)full_stride * (n + 1));
        if (rc) return rc;
        gfull = (uint8_t *)c->gray_full.p;
        const bool prev_needed = has_prev0 && (want_m || (!resized && want_t));
        prof_scope ps_(c, VQA_K_GRAY_HIST);
        if (prev_needed)
            launch_bgr2gray_hist(st, dprev, 1, h, w, frame_stride, row_stride, gfull, gp, full_stride, nullptr, false,
                                 false, false);
        launch_bgr2gray_hist(st, dframes, n, h, w, frame_stride, row_stride, gfull + full_stride, gp, full_stride, res,
                             !resized && want_gh, !resized && want_ch, !resized && want_dct);
    }
    if (need_planes) {
        resize_tabs T;
        rc = get_tabs(c, h, w, rh, rw, T);
        if (rc) return rc;
        rc = ensure(c, c->planeA, (size_t)plane_stride * (n + 1));
        if (rc) return rc;
        rc = ensure(c, c->planeB, (size_t)plane_stride * n);
        if (rc) return rc;
        pA = (uint8_t *)c->planeA.p;
        pB = (uint8_t *)c->planeB.p;
        prof_scope ps_(c, VQA_K_RESIZE);
        if (has_prev0 && want_t)
            launch_resize_planes(st, dprev, 1, h, w, frame_stride, row_stride, rw, rh, T.xofs, T.xa, T.yofs, T.yb,
                                 T.mode, pA, nullptr, pp, plane_stride, nullptr, false, false, false);
        launch_resize_planes(st, dframes, n, h, w, frame_stride, row_stride, rw, rh, T.xofs, T.xa, T.yofs, T.yb, T.mode,
                             pA + plane_stride, pB, pp, plane_stride, res, want_gh, want_ch, want_dct);
    } else if (!resized) {
        pA = gfull;                // slot 0 = prev0
        pB = gfull + full_stride;  // batch frames only
    }

    // ---- VQA_OPT_OVERLAP (default on): block-SAD, the Canny chain, the DCT metrics and ORB only share the gray planes as
    // input, so they run side by side: SAD, Canny and the full-frame DCT (the long one of the two DCT forms) fork onto their
    // own streams here - the planes exist in stream order - and join before the results are copied (QSAD-bound,
    // scalar-bound and latency-bound kernels next to each other: +3 % on the full suite; the full-frame DCT beside the
    // Farneback pyramid: c3ref +9 %, LAB_NOTES.md)
    int dct_mode_eff = P.dct_mode;
    if (dct_mode_eff == VQA_DCT_AUTO) dct_mode_eff = ((int64_t)ph * pw <= 128 * 128) ? VQA_DCT_FULL : VQA_DCT_BLOCK8;
    const bool use_side[3] = {c->opt_overlap && want_m && P.motion_mode == VQA_MOTION_SAD, c->opt_overlap && want_e,
                              c->opt_overlap && (want_dct || want_t) && dct_mode_eff == VQA_DCT_FULL && (want_m || want_e)};
    const bool overlap = use_side[0] || use_side[1] || use_side[2];
    hipStream_t st_sad = st, st_canny = st, st_dct = st;
    if (overlap) {
        for (int i = 0; i < 3; i++) {
            if (!use_side[i]) continue;
            if (!c->side[i]) HIPCHK(c, hipStreamCreateWithFlags(&c->side[i], hipStreamNonBlocking));
            if (!c->ev_join[i]) HIPCHK(c, hipEventCreateWithFlags(&c->ev_join[i], hipEventDisableTiming));
        }
        if (!c->ev_fork) HIPCHK(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
        HIPCHK(c, hipEventRecord(c->ev_fork, st));
        if (use_side[0]) { st_sad = c->side[0]; HIPCHK(c, hipStreamWaitEvent(st_sad, c->ev_fork, 0)); }
        if (use_side[1]) { st_canny = c->side[1]; HIPCHK(c, hipStreamWaitEvent(st_canny, c->ev_fork, 0)); }
        if (use_side[2]) { st_dct = c->side[2]; HIPCHK(c, hipStreamWaitEvent(st_dct, c->ev_fork, 0)); }
    }

    // ---- DCT energy / temporal DCT (input: plane A)
    if (want_dct || want_t) {
        const int mode = dct_mode_eff;
        if (mode == VQA_DCT_BLOCK8) {
            const int pb = dct8_blocks_per_frame(ph, pw);
            rc = ensure(c, c->partials, sizeof(double) * 2 * (size_t)pb * n);
            if (rc) return rc;
            prof_scope ps_(c, VQA_K_DCT8);
            launch_dct8(st, pA, pp, plane_stride, n, ph, pw, want_dct, want_t, has_prev0, (double *)c->partials.p, res, c->dct_wave_slots);
        } else {
            if (dct_fft_supported(ph, pw)) {
                // both sides factor into 2, 3, 5: FFT-based row / column passes (k_dct_fft.hip)
                dct_fft_plan Pw, Ph;
                rc = get_fft_plan(c, pw, &Pw);
                if (rc) return rc;
                rc = get_fft_plan(c, ph, &Ph);
                if (rc) return rc;
                const size_t tiles = (size_t)dct_fft_tiles(ph, pw);
                rc = ensure(c, c->dct_scratch, sizeof(float) * 2 * (size_t)ph * pw * n);
                if (rc) return rc;
                rc = ensure(c, c->dct_pe, sizeof(double) * tiles * n);
                if (rc) return rc;
                rc = ensure(c, c->dct_pt, sizeof(double) * tiles * n);
                if (rc) return rc;
                prof_scope ps_(c, VQA_K_DCT_FULL, st_dct);
                launch_dct_full_fft(st_dct, pA, pp, plane_stride, n, ph, pw, Pw, Ph, (float *)c->dct_scratch.p, (double *)c->dct_pe.p,
                                    (double *)c->dct_pt.p, want_dct, want_t, has_prev0, res);
            } else {
                float *cw = nullptr, *ch = nullptr;
                rc = get_dct_matrix(c, pw, &cw);
                if (rc) return rc;
                rc = get_dct_matrix(c, ph, &ch);
                if (rc) return rc;
                const size_t tiles = (size_t)((ph + 15) / 16) * ((pw + 15) / 16);
                rc = ensure(c, c->dct_scratch, sizeof(float) * (size_t)ph * pw * n);
                if (rc) return rc;
                rc = ensure(c, c->dct_pe, sizeof(double) * tiles * n);
                if (rc) return rc;
                rc = ensure(c, c->dct_pt, sizeof(double) * tiles * n);
                if (rc) return rc;
                prof_scope ps_(c, VQA_K_DCT_FULL, st_dct);
                launch_dct_full(st_dct, pA, pp, plane_stride, n, ph, pw, cw, ch, (float *)c->dct_scratch.p,
                                (double *)c->dct_pe.p, (double *)c->dct_pt.p, want_dct, want_t, has_prev0, res);
            }
        }
    }

    // ---- motion on the full-resolution gray planes (the reference never resizes for it, :327-328):
    // block-SAD (north_star) or the reference's own Farneback flow
    if (want_m && P.motion_mode == VQA_MOTION_SAD) {
        prof_scope ps_(c, VQA_K_SAD, st_sad);
        launch_block_sad(st_sad, gfull, gp, full_stride, n, h, w, P.sad_range, has_prev0, res);
    } else if (want_m) {
        prof_scope ps_(c, VQA_K_FARNEBACK);
        rc = run_farneback(c, st, gfull, gp, full_stride, n, h, w, has_prev0, res);
        if (rc) return rc;
    }

    // ---- Canny (input: plane B)
    c->last_has_state = false;
    if (want_e) {
        const int ww = (pw + 63) / 64;
        const size_t words = (size_t)n * ((ph + 63) / 64) * ww * 64; // tile-major bit-planes
        const unsigned ntiles = canny_hyst_tiles(n, ph, pw);
        rc = ensure(c, c->state, sizeof(uint64_t) * words * 2); // strong plane, then weak plane
        if (rc) return rc;
        rc = ensure(c, c->tile_flags, sizeof(uint32_t) * ntiles * 2); // dedup flags, one array per work list
        if (rc) return rc;
        const size_t NS = CANNY_HYST_SEGMENTS; // list segments (and append counters) per frame
        rc = ensure(c, c->dirty0, sizeof(uint32_t) * ntiles * NS); // work list A
        if (rc) return rc;
        rc = ensure(c, c->dirty1, sizeof(uint32_t) * ntiles * NS); // work list B
        if (rc) return rc;
        rc = ensure(c, c->again_dev, sizeof(uint32_t) * 3 * n * NS); // append counters: three buffers, rotated per round (below)
        if (rc) return rc;
        unsigned long long *strong = (unsigned long long *)c->state.p, *weak = strong + words;
        unsigned *queued[2] = {(unsigned *)c->tile_flags.p, (unsigned *)c->tile_flags.p + ntiles};
        unsigned *lists[2] = {(unsigned *)c->dirty0.p, (unsigned *)c->dirty1.p};
        unsigned *counts = (unsigned *)c->again_dev.p;
        HIPCHK(c, hipMemsetAsync(queued[0], 0, sizeof(uint32_t) * ntiles * 2, st_canny));
        HIPCHK(c, hipMemsetAsync(counts, 0, sizeof(uint32_t) * 3 * n * NS, st_canny));
        // The append counters rotate over THREE buffers so that no memset sits between the rounds: round r consumes
        // cnt(r), appends to cnt(r + 1) and zeroes cnt(r + 2) - the buffer round r - 1 consumed (stream order: done)
        // and round r + 1 will append to.  Lists and dedup flags stay double-buffered (index r & 1).
        auto cnt = [&](int r) { return counts + (size_t)(r % 3) * n * NS; };
        int lo = P.canny_low, hi = P.canny_high;
        if (lo > hi) { int t = lo; lo = hi; hi = t; }
        {
            prof_scope ps_(c, VQA_K_CANNY_NMS, st_canny);
            launch_canny_nms(st_canny, pB, pp, plane_stride, n, ph, pw, lo, hi, strong, weak, res);
        }
        // hysteresis to the fixpoint, entirely enqueued (no host readback).  Round 0 relaxes every tile;
        // round r > 0 relaxes the tiles that round r-1 enqueued into lists[r & 1].
        int round = 0;
        {
            prof_scope ps_(c, VQA_K_CANNY_HYST, st_canny);
            launch_canny_hyst_all(st_canny, strong, weak, n, ph, pw, queued[1], lists[1], cnt(1), res, c->opt_hyst_stats);
        }
        {
            // rounds 1..WIDE (still many tiles): wide grid over the per-frame lists; then the tail kernel
            // finishes every frame on its own workgroup with no host round-trip.
            prof_scope ps_(c, VQA_K_CANNY_HYST, st_canny);
            // (a frame's tail runs on ONE workgroup: big frames in small batches get more wide rounds first)
            const canny_geom cg = canny_tiles(ph, pw);
            int WIDE = (cg.tiles_x * cg.tiles_y > 1024 && n < 256) ? 8 : 6; // (measured: 1080p x 256: 4 -> 0.40 ms, 6 -> 0.38 ms of hysteresis)
#ifdef VQA_AB_VARIANTS
            { static const int e = ab_knob("VQA_HYST_WIDE", 0); if (e > 0) WIDE = e; } // lab build: tuning knob
#endif
            for (round = 1; round <= WIDE; round++) {
                const int in = round & 1, out = in ^ 1;
#ifdef VQA_AB_VARIANTS
                static const bool trace = ab_knob("VQA_HYST_TRACE", 0) != 0;
                if (trace) { // lab build, debugging aid: tiles queued for this round, summed over frames (synchronises)
                    std::vector<uint32_t> hc((size_t)n * NS);
                    (void)hipStreamSynchronize(st_canny);
                    (void)hipMemcpy(hc.data(), cnt(round), sizeof(uint32_t) * hc.size(), hipMemcpyDeviceToHost);
                    unsigned long long tot = 0;
                    for (uint32_t v : hc) tot += v;
                    fprintf(stderr, "[hyst] round %d: %llu of %u tiles queued\n", round, tot, ntiles);
                }
#endif
                launch_canny_hyst_list(st_canny, strong, weak, n, ph, pw, queued[in], lists[in], cnt(round), queued[out], lists[out],
                                       cnt(round + 1), cnt(round + 2), res, c->opt_hyst_stats);
            }
            // the tail alternates between the counter the last wide round appended to and the one it zeroed
            unsigned *tc[2];
            tc[round & 1] = cnt(round);
            tc[(round & 1) ^ 1] = cnt(round + 1);
            launch_canny_hyst_tail(st_canny, strong, weak, n, ph, pw, lists[0], tc[0], queued[0], lists[1], tc[1], queued[1],
                                   round & 1, res, c->opt_hyst_stats, c->seam_hyst_max_rounds, c->seam_hyst_rescue_max_rounds);
        }
        launch_canny_finish(st_canny, strong, n, ph, pw, res);
        c->last_has_state = true;
    }

    // ---- ORB keypoint count: always on the 64x64 thumbnail (:385-386), straight from the BGR frames
    if (want_orb) {
        resize_tabs T64;
        rc = get_tabs(c, h, w, 64, 64, T64);
        if (rc) return rc;
        prof_scope ps_(c, VQA_K_ORB);
        launch_orb64(st, dframes, n, h, w, frame_stride, row_stride, T64.xofs, T64.xa, T64.yofs, T64.yb, T64.mode, res);
    }
    if (overlap) {
        hipStream_t ss[3] = {st_sad, st_canny, st_dct};
        for (int i = 0; i < 3; i++) {
            if (!use_side[i]) continue;
            HIPCHK(c, hipEventRecord(c->ev_join[i], ss[i]));
            HIPCHK(c, hipStreamWaitEvent(st, c->ev_join[i], 0));
        }
    }
    c->last_n = n; c->last_has_full = need_full; c->last_has_planes = need_planes; // (debug planes show the last slice)
    return VQA_OK;
    }; // run_slice
    for (int a0 = 0; a0 < n_all; a0 += SLICE)
        if (const int src = run_slice(a0)) return src; // (the caller drains)
    const bool has_prev0 = batch_has_prev0;

    HIPCHK(c, hipGetLastError());
    if (want_gh || want_ch) {
        HIPCHK(c, hipMemcpyAsync(c->res_host, c->res_dev.p, sizeof(vqa_frame_metrics) * (size_t)n, hipMemcpyDeviceToHost, st));
        c->pend_c_tail_only = false;
    } else {
        // no histogram was asked for: bring back only the 0.6 KB of scalars behind the 4 KB of bins of each record
        const size_t off = offsetof(vqa_frame_metrics, sum_gray2), tail = sizeof(vqa_frame_metrics) - off;
        HIPCHK(c, hipMemcpy2DAsync((uint8_t *)c->res_host + off, sizeof(vqa_frame_metrics), (uint8_t *)c->res_dev.p + off,
                                   sizeof(vqa_frame_metrics), tail, (size_t)n, hipMemcpyDeviceToHost, st));
        c->pend_c_tail_only = true;
    }
    c->pend_c = n;
    c->pend_c_prev0 = has_prev0;
    c->last_h = h; c->last_w = w; c->last_ph = ph; c->last_pw = pw; c->last_pp = pp; c->last_gp = gp;
    c->last_resized = resized;
    return VQA_OK;
}

int vqa_complexity_submit(vqa_ctx *c, const uint8_t *frames, const uint8_t *prev0, int mem_kind, int n, int h, int w,
                          int64_t frame_stride, int64_t row_stride, uint32_t mask, const vqa_params *params)
{
    bool touched = false;
    const int rc = complexity_submit_body(c, frames, prev0, mem_kind, n, h, w, frame_stride, row_stride, mask, params, touched);
    return drain_failed_submit(c, rc, touched);
}

int vqa_complexity_wait(vqa_ctx *c, vqa_frame_metrics *out, int n)
{
    if (!c || !out) return VQA_ERR_INVALID;
    if (!c->pend_c || n != c->pend_c) return VQA_ERR_STATE;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    prof_collect(c);
    memcpy(out, c->res_host, sizeof(vqa_frame_metrics) * (size_t)n);
    for (int i = 0; i < n; i++) {
        if (c->pend_c_tail_only) memset(&out[i], 0, offsetof(vqa_frame_metrics, sum_gray2)); // bins were not computed
        out[i].has_prev = (i > 0 || c->pend_c_prev0) ? 1u : 0u;
    }
    c->pend_c = 0;
    // north_star: edge counts are bit-exact.  A frame whose hysteresis neither the tail nor the rescue pass brought to
    // the fixpoint (hyst_overflow still 1) has a LOWER BOUND in edge_count: that is never handed out as a count.
    for (int i = 0; i < n; i++)
        if (out[i].hyst_overflow == 1u) {
            c->last_err = "Canny hysteresis of frame " + std::to_string(i) + " of " + std::to_string(n) +
                          " stopped before its fixpoint (edge_count " + std::to_string(out[i].edge_count) +
                          " would be a lower bound): records withheld";
            memset(out, 0, sizeof(vqa_frame_metrics) * (size_t)n);
            return VQA_ERR_INCOMPLETE;
        }
    return VQA_OK;
}

// ---------------------------------------------------------------------------
static int quality_submit_body(vqa_ctx *c, const uint8_t *ref, const uint8_t *dist, int mem_kind, int n, int64_t ref_fs,
                               int64_t dist_fs, const vqa_plane_desc *planes, int n_planes, int ssim_mode, bool &touched)
{
    if (!c || !ref || !dist || n <= 0 || !planes || n_planes <= 0 || n_planes > 4) return VQA_ERR_INVALID;
    if (mem_kind != VQA_MEM_HOST && mem_kind != VQA_MEM_DEVICE) return VQA_ERR_INVALID;
    if (ssim_mode != VQA_SSIM_GAUSS && ssim_mode != VQA_SSIM_FFMPEG) return VQA_ERR_INVALID;
    if (c->pend_q) return VQA_ERR_STATE;
    int64_t span = 0;
    int maxblocks = 1;
    for (int p = 0; p < n_planes; p++) {
        const vqa_plane_desc &d = planes[p];
        if (d.width <= 0 || d.height <= 0 || d.offset < 0 || d.pixel_step <= 0 ||
            d.row_stride < (int64_t)d.width * d.pixel_step - (d.pixel_step - 1))
            return VQA_ERR_INVALID;
        // k_ssim_gauss addresses a strip's rows through a 32-bit scalar buffer offset (row * row_stride) plus a 32-bit
        // lane offset: a plane whose rows span 2 GiB (absurd strides / regions of interest only) would wrap silently
        if (ssim_mode == VQA_SSIM_GAUSS &&
            (int64_t)d.height * d.row_stride + (int64_t)d.width * d.pixel_step >= ((int64_t)1 << 31)) return VQA_ERR_UNSUPPORTED;
        if (ssim_mode == VQA_SSIM_GAUSS && (d.width < 11 || d.height < 11)) return VQA_ERR_UNSUPPORTED;
        if (ssim_mode == VQA_SSIM_FFMPEG && (d.width < 8 || d.height < 8)) return VQA_ERR_UNSUPPORTED;
        const int64_t end = d.offset + (int64_t)(d.height - 1) * d.row_stride + (int64_t)(d.width - 1) * d.pixel_step + 1;
        span = end > span ? end : span;
        const int b = ssim_mode == VQA_SSIM_GAUSS ? ssim_gauss_blocks(d.height, d.width) : ssim_ffmpeg_blocks(d.height, d.width);
        maxblocks = b > maxblocks ? b : maxblocks;
    }
    if (n > 1 && (ref_fs < span || dist_fs < span)) return VQA_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    touched = true;
    const uint8_t *dref = ref, *ddist = dist;
    if (mem_kind == VQA_MEM_HOST) {
        const size_t rs = (size_t)(n - 1) * ref_fs + span, ds = (size_t)(n - 1) * dist_fs + span;
        int rc = ensure(c, c->qstage_ref, rs);
        if (rc) return rc;
        rc = ensure(c, c->qstage_dist, ds);
        if (rc) return rc;
        HIPCHK(c, hipMemcpyAsync(c->qstage_ref.p, ref, rs, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync(c->qstage_dist.p, dist, ds, hipMemcpyHostToDevice, st));
        dref = (const uint8_t *)c->qstage_ref.p;
        ddist = (const uint8_t *)c->qstage_dist.p;
    }
    const size_t nent = (size_t)n * n_planes;
    int rc = ensure(c, c->qres_dev, sizeof(vqa_plane_metrics) * nent);
    if (rc) return rc;
    rc = ensure_pinned(c, c->qres_host, c->qres_host_cap, sizeof(vqa_plane_metrics) * nent);
    if (rc) return rc;
    const int QSLICE = 32768; // frames ride in gridDim.y (<= 65535): larger batches go out as consecutive slices
    const int nslice = n < QSLICE ? n : QSLICE;
    rc = ensure(c, c->qpartials, sizeof(double) * (size_t)maxblocks * nslice * n_planes);
    if (rc) return rc;
    HIPCHK(c, hipMemsetAsync(c->qres_dev.p, 0, sizeof(vqa_plane_metrics) * nent, st));
    const int64_t pstride = (int64_t)maxblocks * nslice;
    for (int a0 = 0; a0 < n; a0 += QSLICE) {
        const int m = n - a0 < QSLICE ? n - a0 : QSLICE;
        vqa_plane_metrics *res = (vqa_plane_metrics *)c->qres_dev.p + (size_t)a0 * n_planes;
        const uint8_t *sref = dref + (int64_t)a0 * ref_fs, *sdist = ddist + (int64_t)a0 * dist_fs;
        // planes of identical geometry (B,G,R of packed BGR; U,V of 4:2:0) go out as one group
        bool done[4] = {false, false, false, false};
        for (int p = 0; p < n_planes; p++) {
            if (done[p]) continue;
            int idx[4], cnt = 0;
            for (int q = p; q < n_planes; q++) {
                if (!done[q] && planes[q].width == planes[p].width && planes[q].height == planes[p].height &&
                    planes[q].row_stride == planes[p].row_stride && planes[q].pixel_step == planes[p].pixel_step) {
                    idx[cnt++] = q;
                    done[q] = true;
                }
            }
            prof_scope ps_(c, ssim_mode == VQA_SSIM_GAUSS ? VQA_K_SSIM_GAUSS : VQA_K_SSIM_FFMPEG);
            if (ssim_mode == VQA_SSIM_GAUSS)
                launch_quality_gauss(st, sref, sdist, m, ref_fs, dist_fs, planes, idx, cnt, n_planes,
                                     (double *)c->qpartials.p, pstride, res);
            else
                launch_quality_ffmpeg(st, sref, sdist, m, ref_fs, dist_fs, planes, idx, cnt, n_planes,
                                      (double *)c->qpartials.p, pstride, res);
        }
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(c->qres_host, c->qres_dev.p, sizeof(vqa_plane_metrics) * nent, hipMemcpyDeviceToHost, st));
    c->pend_q = (int)nent;
    return VQA_OK;
}

int vqa_quality_submit(vqa_ctx *c, const uint8_t *ref, const uint8_t *dist, int mem_kind, int n, int64_t ref_fs,
                       int64_t dist_fs, const vqa_plane_desc *planes, int n_planes, int ssim_mode)
{
    bool touched = false;
    const int rc = quality_submit_body(c, ref, dist, mem_kind, n, ref_fs, dist_fs, planes, n_planes, ssim_mode, touched);
    return drain_failed_submit(c, rc, touched);
}

int vqa_quality_wait(vqa_ctx *c, vqa_plane_metrics *out, int n_entries)
{
    if (!c || !out) return VQA_ERR_INVALID;
    if (!c->pend_q || n_entries != c->pend_q) return VQA_ERR_STATE;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    prof_collect(c);
    memcpy(out, c->qres_host, sizeof(vqa_plane_metrics) * (size_t)n_entries);
    c->pend_q = 0;
    return VQA_OK;
}

// ---------------------------------------------------------------------------
int vqa_profile_enable(vqa_ctx *c, int on)
{
    if (!c) return VQA_ERR_INVALID;
    c->prof_on = on != 0;
    return VQA_OK;
}

int vqa_profile_read(vqa_ctx *c, int id, double *total_ms, int64_t *launches, int reset)
{
    if (!c || id < 0 || id >= VQA_K_COUNT) return VQA_ERR_INVALID;
    if (!c->pend_c && !c->pend_q) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        prof_collect(c);
    }
    if (total_ms) *total_ms = c->prof_ms[id];
    if (launches) *launches = c->prof_n[id];
    if (reset) { c->prof_ms[id] = 0; c->prof_n[id] = 0; }
    return VQA_OK;
}

const char *vqa_kernel_name(int id)
{
    static const char *names[VQA_K_COUNT] = {"k_bgr2gray_hist", "k_resize_planes", "k_dct8", "k_dct_full",
                                             "k_canny_nms", "k_canny_hyst", "k_block_sad", "k_ssim_gauss",
                                             "k_ssim_ffmpeg", "k_orb64", "farneback(pyramid)"};
    return (id >= 0 && id < VQA_K_COUNT) ? names[id] : "?";
}

// ---------------------------------------------------------------------------
int vqa_debug_read_plane(vqa_ctx *c, int which, int frame, uint8_t *dst, int dst_h, int dst_w)
{
    if (!c || !dst || frame < 0 || frame >= c->last_n) return VQA_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    const uint8_t *src = nullptr;
    int h = 0, w = 0, pitch = 0;
    if (which == 3) {
        if (!c->last_has_full) return VQA_ERR_STATE;
        h = c->last_h; w = c->last_w; pitch = c->last_gp;
        src = (const uint8_t *)c->gray_full.p + (int64_t)(frame + 1) * h * pitch;
    } else if (which == 0 || which == 1) {
        h = c->last_ph; w = c->last_pw; pitch = c->last_pp;
        if (c->last_resized) {
            if (!c->last_has_planes) return VQA_ERR_STATE;
            src = which == 0 ? (const uint8_t *)c->planeA.p + (int64_t)(frame + 1) * h * pitch
                             : (const uint8_t *)c->planeB.p + (int64_t)frame * h * pitch;
        } else {
            if (!c->last_has_full) return VQA_ERR_STATE;
            src = (const uint8_t *)c->gray_full.p + (int64_t)(frame + 1) * h * pitch;
        }
    } else if (which == 2) {
        if (!c->last_has_state) return VQA_ERR_STATE;
        h = c->last_ph; w = c->last_pw;
        if (dst_h != h || dst_w != w) return VQA_ERR_INVALID;
        const int ww = (w + 63) / 64, ty_n = (h + 63) / 64;
        std::vector<uint64_t> bits((size_t)ty_n * ww * 64);
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipMemcpy(bits.data(), (const uint64_t *)c->state.p + (size_t)frame * bits.size(),
                            sizeof(uint64_t) * bits.size(), hipMemcpyDeviceToHost));
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                const uint64_t wd = bits[(((size_t)(y >> 6) * ww + (x >> 6)) << 6) + (y & 63)];
                dst[(size_t)y * w + x] = ((wd >> (x & 63)) & 1ull) ? 255 : 0;
            }
        return VQA_OK;
    } else {
        return VQA_ERR_INVALID;
    }
    if (dst_h != h || dst_w != w) return VQA_ERR_INVALID;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy2D(dst, (size_t)w, src, (size_t)pitch, (size_t)w, (size_t)h, hipMemcpyDeviceToHost));
    return VQA_OK;
}

} // extern "C"
