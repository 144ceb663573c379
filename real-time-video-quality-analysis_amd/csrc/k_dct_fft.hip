// k_dct_fft.hip — the reference's full-frame cv2.dct metrics at native resolution, the fast way.
//
// Reference functions replaced (complexity_metrics.py):
//   :363-364  np.sum(cv2.dct(np.float32(gray)) ** 2)                      -> dct_energy
//   :574-579  np.sum(np.abs(cv2.dct(prev) - cv2.dct(curr)))  (full frame) -> temporal_dct_l1
//
// "full-frame DCT via mixed-radix (1080 = 2^3 3^3 5, 1920 = 2^7 3 5) row/column passes".
// k_dct_full.hip computes Y = C_h X C_w^T as two dense products (12.4 GFLOP per 1080p frame pair, on the matrix
// cores); this file computes the same orthonormal 2-D DCT-II in O(P log P): every 1-D DCT is ONE complex FFT of the
// same length (Makhoul's reordering v[n] = x[2n], v[N-1-n] = x[2n+1]; C[k] = Re(e^{-i pi k / 2N} V[k])), and TWO real
// sequences - two adjacent rows, then two adjacent columns, of the same plane - ride in the real and imaginary parts of
// one FFT (A[k] = (Z[k] + conj Z[N-k]) / 2, B[k] = (Z[k] - conj Z[N-k]) / 2i).  Two planes are transformed: the current
// plane for the energy, prev - curr for the temporal L1 (linearity: dct(prev) - dct(curr) = dct(prev - curr)).
//
//   k_dct_fft_rows: one workgroup transforms rows: u8 samples -> LDS (centred at 128: the constant's DC term is put
//                   back analytically, which keeps the FFT's rounding noise relative to the texture, not to the
//                   offset) -> Stockham autosort passes of radix 8 / 4 / 2 / 3 / 5 in LDS, IN PLACE (registers hold a
//                   pass between its reads and its writes: half the LDS, 8 workgroups per CU) -> two float planes.
//   k_dct_fft_cols: one workgroup transforms a pair of adjacent columns of both planes the same way and reduces
//                   sum Ya^2 and sum |Yb| on the fly: the 2-D coefficients are never written.
// Lengths must be even and factor into 2, 3, 5 (1080p, 2160p, 720p, 480p ... do); other sizes take k_dct_full.hip.
// No dense contraction is left, so nothing here uses MFMA.  Roofline: HBM (P bytes in + 8P out, 8P in per frame);
// measured at 1080p, 64 frames: rows 0.79 ms + columns 1.5 ms against 15.1 ms for the matrix-core version (first FFT
// version: 1.10 + 1.96) - the row pass is bound by the barriers between its short radix passes (latency / occupancy),
// the column pass by its strided reads, neither by HBM.
#include "vqa_dev.hpp"
#include "vqa_kernels.hpp"

namespace vqa {

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 mulmi(float2 a) { return make_float2(a.y, -a.x); } // a * (-i)

// forward DFTs of length R (kernel e^{-2 pi i / R}), in place
template <int R>
__device__ __forceinline__ void dft(float2 *v);
template <>
__device__ __forceinline__ void dft<2>(float2 *v)
{
    const float2 a = v[0], b = v[1];
    v[0] = cadd(a, b); v[1] = csub(a, b);
}
template <>
__device__ __forceinline__ void dft<3>(float2 *v)
{
    const float2 s = cadd(v[1], v[2]), d = csub(v[1], v[2]);
    const float2 m = make_float2(v[0].x - 0.5f * s.x, v[0].y - 0.5f * s.y);
    const float2 r = make_float2(0.86602540378443865f * d.y, -0.86602540378443865f * d.x); // (-i sqrt3/2) d
    v[0] = cadd(v[0], s); v[1] = cadd(m, r); v[2] = csub(m, r);
}
template <>
__device__ __forceinline__ void dft<4>(float2 *v)
{
    const float2 t0 = cadd(v[0], v[2]), t1 = csub(v[0], v[2]), t2 = cadd(v[1], v[3]), t3 = mulmi(csub(v[1], v[3]));
    v[0] = cadd(t0, t2); v[1] = cadd(t1, t3); v[2] = csub(t0, t2); v[3] = csub(t1, t3);
}
template <>
__device__ __forceinline__ void dft<8>(float2 *v)
{
    // two radix-4 transforms of the even / odd samples, combined with the eighth roots of unity
    float2 e[4] = {v[0], v[2], v[4], v[6]}, o[4] = {v[1], v[3], v[5], v[7]};
    dft<4>(e);
    dft<4>(o);
    constexpr float r = 0.70710678118654752f;
    const float2 o1 = make_float2(r * (o[1].x + o[1].y), r * (o[1].y - o[1].x));   // o1 (1 - i) / sqrt2
    const float2 o2 = mulmi(o[2]);                                                 // o2 (-i)
    const float2 o3 = make_float2(r * (o[3].y - o[3].x), -r * (o[3].x + o[3].y));  // o3 (-1 - i) / sqrt2
    v[0] = cadd(e[0], o[0]); v[4] = csub(e[0], o[0]);
    v[1] = cadd(e[1], o1);   v[5] = csub(e[1], o1);
    v[2] = cadd(e[2], o2);   v[6] = csub(e[2], o2);
    v[3] = cadd(e[3], o3);   v[7] = csub(e[3], o3);
}
template <>
__device__ __forceinline__ void dft<5>(float2 *v)
{
    constexpr float c1 = 0.30901699437494742f, c2 = -0.80901699437494742f, s1 = 0.95105651629515357f, s2 = 0.58778525229247313f;
    const float2 a1 = cadd(v[1], v[4]), a2 = cadd(v[2], v[3]), b1 = csub(v[1], v[4]), b2 = csub(v[2], v[3]);
    const float2 m1 = make_float2(v[0].x + c1 * a1.x + c2 * a2.x, v[0].y + c1 * a1.y + c2 * a2.y);
    const float2 m2 = make_float2(v[0].x + c2 * a1.x + c1 * a2.x, v[0].y + c2 * a1.y + c1 * a2.y);
    const float2 n1 = mulmi(make_float2(s1 * b1.x + s2 * b2.x, s1 * b1.y + s2 * b2.y));
    const float2 n2 = mulmi(make_float2(s2 * b1.x - s1 * b2.x, s2 * b1.y - s1 * b2.y));
    v[0] = make_float2(v[0].x + a1.x + a2.x, v[0].y + a1.y + a2.y);
    v[1] = cadd(m1, n1); v[4] = csub(m1, n1); v[2] = cadd(m2, n2); v[3] = csub(m2, n2);
}

// One Stockham autosort pass of radix R over `cols` independent sequences of length N stored back to back:
// butterfly j of a sequence reads in[j + t N/R], twiddles input t by e^{-2 pi i t k / (Ns R)} (k = j mod Ns, Ns = the
// product of the radices already applied) and writes out[(j - k) R + k + t Ns].  tw[m] = e^{-2 pi i m / N}.
// (No integer division in the loop: the GPU has none in hardware and two of them doubled the instruction count of a
// butterfly.  M and tstep come with the plan; j mod Ns is a multiply-high by ceil(2^32 / Ns), exact while j Ns < 2^32.)
// IN PLACE: every butterfly of the pass is read (and transformed) into registers, a barrier, then written back to the
// SAME buffer - two barriers per pass instead of one, but half the LDS per sequence, and these passes live on occupancy
// (a phase gives a thread about one butterfly between barriers; round 4: 5 -> 8 workgroups per CU for the 1080p rows).
// A thread holds ceil(M / NT) butterflies: at most MAXN / (R * NT) of R points, MAXN / NT complex values in all; the size
// class above 2048 points runs 512 threads per workgroup so that the register file of a pass stays the same.
// MAXN = the size class of the transform (2048 or 4096): it sizes the register file of a pass.
template <int R, int MAXN, int NT>
__device__ __forceinline__ void fft_pass(float2 *__restrict__ buf, const float2 *__restrict__ tw, int N, int Ns,
                                         uint32_t ns_magic, int M, int tstep, int cols, int tid)
{
    constexpr int IT = (MAXN / R + NT - 1) / NT;
    for (int c = 0; c < cols; c++) {
        float2 v[IT][R];
        int dsto[IT];
#pragma unroll
        for (int it = 0; it < IT; it++) {
            const int j = tid + it * NT;
            if (j < M) {
                const int k = Ns == 1 ? 0 : j - (int)__umulhi((uint32_t)j, ns_magic) * Ns; // j mod Ns  (Ns == 1: wave-uniform)
                const float2 *src = buf + c * N + j;
                v[it][0] = src[0];
#pragma unroll
                for (int t = 1; t < R; t++) v[it][t] = cmul(src[t * M], tw[t * k * tstep]);
                dft<R>(v[it]);
                dsto[it] = c * N + (j - k) * R + k;
            }
        }
        __syncthreads(); // every butterfly of the pass has been read
#pragma unroll
        for (int it = 0; it < IT; it++) {
            const int j = tid + it * NT;
            if (j < M) {
                float2 *dst = buf + dsto[it];
#pragma unroll
                for (int t = 0; t < R; t++) dst[t * Ns] = v[it][t];
            }
        }
        __syncthreads();
    }
}

// all passes of a plan, in place in `buf` (cols sequences of P.n <= MAXN back to back); NT threads
template <int MAXN, int NT>
__device__ __forceinline__ void fft_run(const dct_fft_plan &P, float2 *buf, const float2 *tw, int cols, int tid)
{
    int Ns = 1;
    for (int p = 0; p < P.npass; p++) {
        const int R = P.radix[p];
        const uint32_t mg = P.ns_magic[p];
        const int M = P.m[p], ts = P.tstep[p];
        if (R == 8) fft_pass<8, MAXN, NT>(buf, tw, P.n, Ns, mg, M, ts, cols, tid);
        else if (R == 4) fft_pass<4, MAXN, NT>(buf, tw, P.n, Ns, mg, M, ts, cols, tid);
        else if (R == 2) fft_pass<2, MAXN, NT>(buf, tw, P.n, Ns, mg, M, ts, cols, tid);
        else if (R == 3) fft_pass<3, MAXN, NT>(buf, tw, P.n, Ns, mg, M, ts, cols, tid);
        else fft_pass<5, MAXN, NT>(buf, tw, P.n, Ns, mg, M, ts, cols, tid);
        Ns *= R;
    }
}

// The column pass keeps the OUT-OF-PLACE form (two buffers, one barrier per pass): it is bound by its strided global
// reads, not by occupancy, and the second barrier of the in-place form cost it 2-17 % (720p 0.59 -> 0.70 ms, 2160p 5.5 -> 6.4).
template <int R>
__device__ __forceinline__ void fft_pass_oop(const float2 *__restrict__ in, float2 *__restrict__ out, const float2 *__restrict__ tw,
                                         int N, int Ns, uint32_t ns_magic, int M, int tstep, int cols, int tid, int nthreads)
{
    for (int c = 0; c < cols; c++)
    for (int j = tid; j < M; j += nthreads) {
        const int k = Ns == 1 ? 0 : j - (int)__umulhi((uint32_t)j, ns_magic) * Ns; // j mod Ns  (Ns == 1: wave-uniform)
        const float2 *src = in + c * N + j;
        float2 v[R];
        v[0] = src[0];
#pragma unroll
        for (int t = 1; t < R; t++) v[t] = cmul(src[t * M], tw[t * k * tstep]);
        dft<R>(v);
        float2 *dst = out + c * N + (j - k) * R + k;
#pragma unroll
        for (int t = 0; t < R; t++) dst[t * Ns] = v[t];
    }
}

// all passes of a plan; returns the buffer that holds the (naturally ordered) spectrum
__device__ __forceinline__ float2 *fft_run_oop(const dct_fft_plan &P, float2 *b0, float2 *b1, const float2 *tw, int cols, int tid,
                                           int nthreads)
{
    float2 *in = b0, *out = b1;
    int Ns = 1;
    for (int p = 0; p < P.npass; p++) {
        const int R = P.radix[p];
        const uint32_t mg = P.ns_magic[p];
        const int M = P.m[p], ts = P.tstep[p];
        if (R == 8) fft_pass_oop<8>(in, out, tw, P.n, Ns, mg, M, ts, cols, tid, nthreads);
        else if (R == 4) fft_pass_oop<4>(in, out, tw, P.n, Ns, mg, M, ts, cols, tid, nthreads);
        else if (R == 2) fft_pass_oop<2>(in, out, tw, P.n, Ns, mg, M, ts, cols, tid, nthreads);
        else if (R == 3) fft_pass_oop<3>(in, out, tw, P.n, Ns, mg, M, ts, cols, tid, nthreads);
        else fft_pass_oop<5>(in, out, tw, P.n, Ns, mg, M, ts, cols, tid, nthreads);
        __syncthreads();
        float2 *t = in; in = out; out = t;
        Ns *= R;
    }
    return in;
}

// Makhoul's input order: even samples ascending, odd samples descending
__device__ __forceinline__ int makhoul_pos(int n, int N) { return (n & 1) ? N - 1 - (n >> 1) : (n >> 1); }

// the two real spectra packed in Z, then the DCT-II coefficients of both: post[k] = s_k (cos, sin)(pi k / 2N)
__device__ __forceinline__ float2 dct_from_fft(const float2 *Z, int k, int N, float2 post)
{
    const float2 zk = Z[k], zn = Z[k ? N - k : 0];
    const float ax = 0.5f * (zk.x + zn.x), ay = 0.5f * (zk.y - zn.y); // A = (Z[k] + conj Z[N-k]) / 2
    const float bx = 0.5f * (zk.y + zn.y), by = -0.5f * (zk.x - zn.x); // B = (Z[k] - conj Z[N-k]) / 2i
    return make_float2(post.x * ax + post.y * ay, post.x * bx + post.y * by);
}

// grid = (ceil(h / rows_per_wg), n_frames), block = 256, dynamic LDS = NSEQ * w * 8 bytes (the passes run in place); rows_per_wg is even.
// One FFT transforms a PAIR of rows of the SAME plane (row r in the real part, row r + 1 in the imaginary part); with
// NSEQ = 2 the pair of the plane and the pair of prev - curr go through the passes together (half the barriers per row).
template <int NSEQ, int MAXN, int NT>
__global__ __launch_bounds__(NT) void k_dct_fft_rows(const uint8_t *__restrict__ planes, int pitch, int64_t plane_stride, int h,
                                                      int w, dct_fft_plan P, float *__restrict__ Ra, float *__restrict__ Rb,
                                                      int rows_per_wg, int want_a, int want_b)
{
    extern __shared__ float2 lds_fft[];
    float2 *b0 = lds_fft;
    // twiddles come from the plan's table in global memory (L1-resident): a copy in LDS cost 15 KB per workgroup, i.e. two of
    // the five workgroups a CU can hold, and these passes live on occupancy (1.00 ms with the copy, 0.90 without, 1080p)
    const float2 *tw = P.tw;
    const int tid = threadIdx.x, f = blockIdx.y;
    const uint8_t *cur = planes + (int64_t)(f + 1) * plane_stride, *prev = planes + (int64_t)f * plane_stride;
    const float dc = 128.f * sqrtf((float)w); // row-DCT of the constant that was subtracted
    const int r0 = blockIdx.x * rows_per_wg, r1 = min(h, r0 + rows_per_wg);
    int which[2], np_ = 0; // 0: the plane itself (centred), 1: prev - curr
    if (want_a) which[np_++] = 0;
    if (want_b) which[np_++] = 1;
    for (int r = r0; r < r1; r += 2) {          // h is even
        for (int p0 = 0; p0 < np_; p0 += NSEQ) {
            const int ns = min(NSEQ, np_ - p0);
            __syncthreads(); // the previous transform's readers are done
            for (int q = 0; q < ns; q++) {
                const int pl = which[p0 + q];
                float2 *dst = b0 + q * w;
                for (int n = tid; n < w; n += NT) {
                    const int c0 = cur[(int64_t)r * pitch + n], c1 = cur[(int64_t)(r + 1) * pitch + n];
                    float2 v;
                    if (pl == 0) v = make_float2((float)(c0 - 128), (float)(c1 - 128));
                    else v = make_float2((float)((int)prev[(int64_t)r * pitch + n] - c0), (float)((int)prev[(int64_t)(r + 1) * pitch + n] - c1));
                    dst[makhoul_pos(n, w)] = v;
                }
            }
            __syncthreads();
            fft_run<MAXN, NT>(P, b0, tw, ns, tid); // ONE call site for every mask
            const float2 *Z = b0;
            for (int q = 0; q < ns; q++) {
                const int pl = which[p0 + q];
                float *o0 = (pl ? Rb : Ra) + ((int64_t)f * h + r) * w, *o1 = o0 + w;
                const float add = pl ? 0.f : dc;
                for (int k = tid; k < w; k += NT) {
                    const float2 c = dct_from_fft(Z + q * w, k, w, P.post[k]);
                    o0[k] = k ? c.x : c.x + add;
                    o1[k] = k ? c.y : c.y + add;
                }
            }
        }
    }
}

// grid = (8 * ceil(w / 2 / 8), n_frames), block = 256.  A workgroup takes the column pair (x0, x0 + 1) of both planes:
// two FFTs (one per plane; column x0 in the real part, x0 + 1 in the imaginary part), CONCURRENT when the LDS holds both
// (dynamic LDS = 4 * h * 8 bytes), else one after the other (2 * h * 8).  sum Ya^2 and sum |Yb| are reduced
// on the fly: the 2-D coefficients are never written.
template <bool BOTH>
__global__ __launch_bounds__(256) void k_dct_fft_cols(const float *__restrict__ Ra, const float *__restrict__ Rb, int h, int w,
                                                      dct_fft_plan P, double *__restrict__ pe, double *__restrict__ pt, int want_a,
                                                      int want_b)
{
    extern __shared__ float2 lds_fft[];
    __shared__ double red[4];
    constexpr int NSEQ = BOTH ? 2 : 1;
    float2 *b0 = lds_fft, *b1 = lds_fft + NSEQ * h;
    const float2 *twl = P.tw; // (global / L1, as in the row pass)
    const int tid = threadIdx.x, f = blockIdx.y;
    // XCD-aware tile order (workgroup ids go round-robin over the 8 XCDs): XCD c takes a contiguous range of column pairs,
    // so the neighbours that share 64-byte sectors of a row meet in ONE L2
    const int per = ((int)gridDim.x + 7) >> 3;
    const int tile = (int)(blockIdx.x & 7u) * per + (int)(blockIdx.x >> 3);
    const int ntiles = w >> 1;
    if ((int)(blockIdx.x >> 3) >= per || tile >= ntiles) return;
    const int x0 = tile * 2;
    const float *pa = Ra + (int64_t)f * h * w + x0, *pb = Rb + (int64_t)f * h * w + x0;
    double e = 0, t = 0;
    auto load = [&](const float *src, float2 *dst) {
        for (int m = tid; m < h; m += 256) dst[makhoul_pos(m, h)] = *(const float2 *)(src + (int64_t)m * w); // x0 is even: 8-byte aligned
    };
    auto reduce = [&](const float2 *Z, bool energy) {
        for (int k = tid; k < h; k += 256) {
            const float2 y = dct_from_fft(Z, k, h, P.post[k]);
            if (energy) e += (double)y.x * (double)y.x + (double)y.y * (double)y.y;
            else t += (double)fabsf(y.x) + (double)fabsf(y.y);
        }
    };
    // ONE call site of the transform for every mask (energy only, temporal only, both): the same inlined arithmetic, so a
    // metric's bits do not depend on which other metric was asked for
    int which[2], np_ = 0;
    if (want_a) which[np_++] = 0;
    if (want_b) which[np_++] = 1;
    for (int p0 = 0; p0 < np_; p0 += NSEQ) {
        const int ns = min(NSEQ, np_ - p0);
        __syncthreads(); // the previous round's readers are done
        for (int q = 0; q < ns; q++) load(which[p0 + q] ? pb : pa, b0 + q * h);
        __syncthreads();
        const float2 *Z = fft_run_oop(P, b0, b1, twl, ns, tid, 256);
        for (int q = 0; q < ns; q++) reduce(Z + q * h, which[p0 + q] == 0);
    }
    const double es = block_sum(e, red);
    const double ts = block_sum(t, red);
    if (tid == 0) {
        pe[(int64_t)f * ntiles + tile] = es;
        pt[(int64_t)f * ntiles + tile] = ts;
    }
}

// ---- column pass, round 4 form: EIGHT columns of one plane per workgroup ------------------------------------------
// The pair-per-workgroup kernel above reads 8 bytes per row at a stride of a whole row: every lane of a load touches its
// own cache line.  Here a workgroup takes the 32-byte segment (four column pairs) of every row of ONE plane into an LDS
// slab z[makhoul(row)][pair] (float2: column 2p in the real part, 2p + 1 in the imaginary part), runs the radix passes on
// the four interleaved sequences IN PLACE (a pass's butterflies of all four sequences go through registers between two
// barriers: a quarter of the barriers per sequence, and consecutive lanes touch consecutive LDS words), and reduces
// sum Y^2 (plane a) or sum |Y| (plane b) straight from the slab.  One plane per workgroup (blockIdx.z), so a metric's bits
// cannot depend on whether the other one was asked for.  LDS = 32 h bytes (1080p: 34.5 KB, four workgroups per CU).
// grid = (8 * ceil(ceil(w / 8) / 8), n_frames, planes wanted), block = NT; MAXN = size class (register file of a pass).
// Used up to 2040 rows (64 KiB of LDS); taller planes keep the pair-per-workgroup kernel above.
template <int R, int MAXN, int NT, int NS>
__device__ __forceinline__ void fft_passN(float2 *__restrict__ z, const float2 *__restrict__ tw, int Ns, uint32_t ns_magic, int M,
                                          int tstep, int tid)
{
    static_assert(NS == 2 || NS == 4, "interleaved sequences");
    constexpr int IT = (NS * (MAXN / R) + NT - 1) / NT;
    float2 v[IT][R];
    int dsto[IT];
#pragma unroll
    for (int it = 0; it < IT; it++) {
        const int q = tid + it * NT, c = q & (NS - 1), j = q / NS;
        if (j < M) {
            const int k = Ns == 1 ? 0 : j - (int)__umulhi((uint32_t)j, ns_magic) * Ns;
            const float2 *src = z + j * NS + c;
            v[it][0] = src[0];
#pragma unroll
            for (int t = 1; t < R; t++) v[it][t] = cmul(src[t * M * NS], tw[t * k * tstep]);
            dft<R>(v[it]);
            dsto[it] = ((j - k) * R + k) * NS + c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < IT; it++) {
        const int q = tid + it * NT, j = q / NS;
        if (j < M) {
            float2 *dst = z + dsto[it];
#pragma unroll
            for (int t = 0; t < R; t++) dst[t * Ns * NS] = v[it][t];
        }
    }
    __syncthreads();
}

template <int MAXN, int NT>
__global__ __launch_bounds__(NT) void k_dct_fft_cols8(const float *__restrict__ Ra, const float *__restrict__ Rb, int h, int w,
                                                       dct_fft_plan P, double *__restrict__ pe, double *__restrict__ pt, int want_a,
                                                       int want_b, int ntiles)
{
    extern __shared__ float2 lds_fft[];
    __shared__ double red[NT / 64];
    float2 *z = lds_fft;
    const float2 *tw = P.tw;
    const int tid = threadIdx.x, f = blockIdx.y;
    const bool plane_b = want_a ? blockIdx.z == 1 : true; // z = 0: the first wanted plane
    (void)want_b;
    // XCD-aware tile order: XCD c takes a contiguous range of tiles, so the two tiles that share a 64-byte sector meet in ONE L2
    const int per = ((int)gridDim.x + 7) >> 3;
    const int tile = (int)(blockIdx.x & 7u) * per + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per || tile >= ntiles) return;
    const int x0 = tile * 8, npairs = min(4, (w - x0) >> 1); // w is even
    const float *src = (plane_b ? Rb : Ra) + (int64_t)f * h * w + x0;
    for (int i = tid; i < h * 4; i += NT) {
        const int m = i >> 2, c = i & 3;
        float2 val = make_float2(0.f, 0.f);
        if (c < npairs) val = *(const float2 *)(src + (int64_t)m * w + 2 * c); // x0 + 2c is even: 8-byte aligned
        z[makhoul_pos(m, h) * 4 + c] = val;
    }
    __syncthreads();
    int Ns = 1;
    for (int p = 0; p < P.npass; p++) {
        const int R = P.radix[p];
        const uint32_t mg = P.ns_magic[p];
        const int M = P.m[p], ts = P.tstep[p];
        if (R == 8) fft_passN<8, MAXN, NT, 4>(z, tw, Ns, mg, M, ts, tid);
        else if (R == 4) fft_passN<4, MAXN, NT, 4>(z, tw, Ns, mg, M, ts, tid);
        else if (R == 2) fft_passN<2, MAXN, NT, 4>(z, tw, Ns, mg, M, ts, tid);
        else if (R == 3) fft_passN<3, MAXN, NT, 4>(z, tw, Ns, mg, M, ts, tid);
        else fft_passN<5, MAXN, NT, 4>(z, tw, Ns, mg, M, ts, tid);
        Ns *= R;
    }
    double acc = 0;
    for (int i = tid; i < h * 4; i += NT) {
        const int k = i >> 2, c = i & 3;
        if (c >= npairs) continue;
        const float2 zk = z[k * 4 + c], zn = z[(k ? h - k : 0) * 4 + c], post = P.post[k];
        const float ax = 0.5f * (zk.x + zn.x), ay = 0.5f * (zk.y - zn.y); // as dct_from_fft
        const float bx = 0.5f * (zk.y + zn.y), by = -0.5f * (zk.x - zn.x);
        const float y0 = post.x * ax + post.y * ay, y1 = post.x * bx + post.y * by;
        if (plane_b) acc += (double)fabsf(y0) + (double)fabsf(y1);
        else acc += (double)y0 * (double)y0 + (double)y1 * (double)y1;
    }
    const double tot = block_sum(acc, red);
    if (tid == 0) (plane_b ? pt : pe)[(int64_t)f * ntiles + tile] = tot;
}

#ifndef DCT_ROWS2
#define DCT_ROWS2 0 // 1 = measurement build (scripts/build_probes.sh DCT_ROWS2 1): two row pairs per transform round
#endif
#if DCT_ROWS2
// ---- row pass, two row pairs per transform round --------------------------------------------------------------------
// As k_dct_fft_rows, but TWO row pairs of the same plane (rows r..r+1 and r+2..r+3) go through the passes together as
// two interleaved sequences z[makhoul(n)][pair] (the form of the column pass above): half the barriers per sequence and
// twice the independent butterflies per thread between them.  LDS = 2 w 8 bytes (30 KB at 1920: five workgroups per CU).
// MEASURED SLOWER than one pair per round (1080p 0.92 against 0.79 ms per 64 frames, 720p 0.38 against 0.33: 94 VGPRs and
// 30 KB leave five workgroups per CU where the single-pair kernel has eight): kept as a measurement build only.
// grid = (ceil(h / rows_per_wg), n_frames), block = NT; rows_per_wg a multiple of 4.
template <int MAXN, int NT>
__global__ __launch_bounds__(NT) void k_dct_fft_rows2(const uint8_t *__restrict__ planes, int pitch, int64_t plane_stride, int h,
                                                       int w, dct_fft_plan P, float *__restrict__ Ra, float *__restrict__ Rb,
                                                       int rows_per_wg, int want_a, int want_b)
{
    extern __shared__ float2 lds_fft[];
    float2 *z = lds_fft;
    const float2 *tw = P.tw;
    const int tid = threadIdx.x, f = blockIdx.y;
    const uint8_t *cur = planes + (int64_t)(f + 1) * plane_stride, *prev = planes + (int64_t)f * plane_stride;
    const float dc = 128.f * sqrtf((float)w); // row-DCT of the constant that was subtracted
    const int r0 = blockIdx.x * rows_per_wg, r1 = min(h, r0 + rows_per_wg);
    int which[2], np_ = 0; // 0: the plane itself (centred), 1: prev - curr
    if (want_a) which[np_++] = 0;
    if (want_b) which[np_++] = 1;
    for (int r = r0; r < r1; r += 4) {          // h is even: a round holds one or two row pairs
        const int npair = min(2, (r1 - r) >> 1);
        for (int p0 = 0; p0 < np_; p0++) {
            const int pl = which[p0];
            __syncthreads(); // the previous round's readers are done
            for (int i = tid; i < 2 * w; i += NT) {
                const int q = i >= w, n = i - q * w, row = r + 2 * q;
                float2 v = make_float2(0.f, 0.f);
                if (q < npair) {
                    const int c0 = cur[(int64_t)row * pitch + n], c1 = cur[(int64_t)(row + 1) * pitch + n];
                    if (pl == 0) v = make_float2((float)(c0 - 128), (float)(c1 - 128));
                    else v = make_float2((float)((int)prev[(int64_t)row * pitch + n] - c0), (float)((int)prev[(int64_t)(row + 1) * pitch + n] - c1));
                }
                z[makhoul_pos(n, w) * 2 + q] = v;
            }
            __syncthreads();
            int Ns = 1;
            for (int p = 0; p < P.npass; p++) { // ONE call site of the passes for every mask
                const int R = P.radix[p];
                const uint32_t mg = P.ns_magic[p];
                const int M = P.m[p], ts = P.tstep[p];
                if (R == 8) fft_passN<8, MAXN, NT, 2>(z, tw, Ns, mg, M, ts, tid);
                else if (R == 4) fft_passN<4, MAXN, NT, 2>(z, tw, Ns, mg, M, ts, tid);
                else if (R == 2) fft_passN<2, MAXN, NT, 2>(z, tw, Ns, mg, M, ts, tid);
                else if (R == 3) fft_passN<3, MAXN, NT, 2>(z, tw, Ns, mg, M, ts, tid);
                else fft_passN<5, MAXN, NT, 2>(z, tw, Ns, mg, M, ts, tid);
                Ns *= R;
            }
            const float add = pl ? 0.f : dc;
            for (int i = tid; i < 2 * w; i += NT) {
                const int q = i >= w, k = i - q * w;
                if (q >= npair) continue;
                const float2 zk = z[k * 2 + q], zn = z[(k ? w - k : 0) * 2 + q], post = P.post[k];
                const float ax = 0.5f * (zk.x + zn.x), ay = 0.5f * (zk.y - zn.y); // as dct_from_fft
                const float bx = 0.5f * (zk.y + zn.y), by = -0.5f * (zk.x - zn.x);
                const float y0 = post.x * ax + post.y * ay, y1 = post.x * bx + post.y * by;
                float *o0 = (pl ? Rb : Ra) + ((int64_t)f * h + r + 2 * q) * w;
                o0[k] = k ? y0 : y0 + add;
                o0[w + k] = k ? y1 : y1 + add;
            }
        }
    }
}

#endif // DCT_ROWS2

// ---- host side ---------------------------------------------------------------------------------------------
// radix-8 passes first, then 4, 2, 3, 5; false if n is odd, has another prime factor, or is too short / too long to pay
bool dct_fft_factor(int n, int radix[DCT_FFT_MAX_PASSES], int *npass)
{
    if (n < 128 || n > 4000 || (n & 1)) return false; // (two sequences of n float2 within 64 KiB of LDS; FFT_MAXN registers)
    int m = n, np = 0;
    while (m % 8 == 0 && np < DCT_FFT_MAX_PASSES) { radix[np++] = 8; m /= 8; }
    while (m % 4 == 0 && np < DCT_FFT_MAX_PASSES) { radix[np++] = 4; m /= 4; }
    static const int primes[3] = {2, 3, 5};
    for (int pi = 0; pi < 3; pi++)
        while (m % primes[pi] == 0 && np < DCT_FFT_MAX_PASSES) { radix[np++] = primes[pi]; m /= primes[pi]; }
    if (m != 1) return false;
    *npass = np;
    return true;
}

// Packing history: the first version put the plane in the real part and prev - curr in the imaginary part of ONE
// FFT.  The separation (Z[k] +- conj Z[N-k]) / 2 is exact only in exact arithmetic: rounding noise of the LARGE plane
// leaks into the SMALL difference (~1e-7 of sum |A|: identical frames read 0.5 instead of 0, a static scene with a few
// changed pixels is off by per cents).  Packing two rows / columns of the SAME plane keeps the leak relative to that
// plane's own magnitude, and a zero difference stays exactly zero.
// both sides factor (a transform of up to 4000 points - two buffers of n float2 - stays inside 64 KiB of LDS; 4096 would
// need all 65536 bytes of the default dynamic-LDS limit and is left to the dense path)
bool dct_fft_supported(int h, int w)
{
    int rx[DCT_FFT_MAX_PASSES], np;
    return dct_fft_factor(h, rx, &np) && dct_fft_factor(w, rx, &np);
}

int dct_fft_tiles(int h, int w) { (void)h; return w >> 1; } // one partial pair per column pair

// planes: slot 0 = frame before the batch, slot i+1 = batch frame i (u8, pitch).  scratch: 2 * n * h * w floats.
// pe / pt: n * dct_fft_tiles(h, w) doubles each.  The finalize kernel is k_dct_full.hip's.
void launch_dct_full_fft(hipStream_t st, const uint8_t *planes, int pitch, int64_t plane_stride, int n, int h, int w,
                         const dct_fft_plan &pw_, const dct_fft_plan &ph_, float *scratch, double *pe, double *pt, bool energy,
                         bool temporal, bool first_has_prev, vqa_frame_metrics *res)
{
    if (n <= 0 || (!energy && !temporal)) return;
    float *Ra = scratch, *Rb = scratch + (int64_t)n * h * w;
    const int cap = 64 * 1024, rpw = 8;
    const dim3 gr((h + rpw - 1) / rpw, n);
    // (NSEQ = 2 - both planes' row pairs through the passes together - measured SLOWER at every size: 1080p 1.52 ms against
    // 1.11, 720p 0.45 against 0.37, 540p 0.22 against 0.20: larger LDS footprint, fewer workgroups per CU)
#if DCT_ROWS2
    if (w <= 2048)
        hipLaunchKernelGGL((k_dct_fft_rows2<2048, 256>), gr, dim3(256), (size_t)2 * w * 8, st, planes, pitch, plane_stride, h, w, pw_, Ra,
                           Rb, rpw, (int)energy, (int)temporal);
    else
#endif
    if (w <= 2048)
        hipLaunchKernelGGL((k_dct_fft_rows<1, 2048, 256>), gr, dim3(256), (size_t)w * 8, st, planes, pitch, plane_stride, h, w, pw_, Ra, Rb,
                           rpw, (int)energy, (int)temporal);
    else
        hipLaunchKernelGGL((k_dct_fft_rows<1, 4096, 512>), gr, dim3(512), (size_t)w * 8, st, planes, pitch, plane_stride, h, w, pw_, Ra, Rb,
                           rpw, (int)energy, (int)temporal);
    // eight columns of one plane per workgroup (32 h bytes of LDS)
    int tiles = (w + 7) / 8;
    const dim3 gc((tiles + 7) / 8 * 8, n, (energy ? 1 : 0) + (temporal ? 1 : 0));
    if (h <= 1152) {
        hipLaunchKernelGGL((k_dct_fft_cols8<1152, 256>), gc, dim3(256), (size_t)h * 32, st, Ra, Rb, h, w, ph_, pe, pt, (int)energy,
                           (int)temporal, tiles);
    } else if (h <= 2040) {
        hipLaunchKernelGGL((k_dct_fft_cols8<2048, 256>), gc, dim3(256), (size_t)h * 32, st, Ra, Rb, h, w, ph_, pe, pt, (int)energy,
                           (int)temporal, tiles);
    } else {
        // taller planes: the slab would exceed 64 KiB (2160 rows: 69 KB, ONE 512-thread workgroup per CU - measured 5.8 ms
        // against 5.6 for the pair-per-workgroup kernel, 64 x 2160p): the pair kernel stays
        tiles = w >> 1;
        const dim3 gp((tiles + 7) / 8 * 8, n);
        if (4 * h * 8 <= cap - 64)
            hipLaunchKernelGGL(k_dct_fft_cols<true>, gp, dim3(256), (size_t)4 * h * 8, st, Ra, Rb, h, w, ph_, pe, pt, (int)energy, (int)temporal);
        else
            hipLaunchKernelGGL(k_dct_fft_cols<false>, gp, dim3(256), (size_t)2 * h * 8, st, Ra, Rb, h, w, ph_, pe, pt, (int)energy, (int)temporal);
    }
    launch_dct_full_finalize(st, pe, pt, tiles, n, res, energy, temporal, first_has_prev);
}

} // namespace vqa
