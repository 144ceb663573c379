// k_dct_full.hip — full-frame orthonormal 2-D DCT-II, i.e. exactly what the
// reference's cv2.dct computes (complexity_metrics.py:363, :574-575), for the
// reference's own configuration (config.json: 64x64 after resize).
//
//   Y = C_h X C_w^T  as two NT products in fp32 on the vector ALUs:
//     Tt = C_w . X^T         (w x h)      X = gray, or prev - curr (linearity)
//     Yt = Tt . C_h^T        (w x h)      reduced on the fly: sum Y^2 or sum |Y|
//
// Planes whose sides are even and factor into 2, 3, 5 (1080p, 2160p, 720p ...) take k_dct_fft.hip instead (round 4:
// the same metrics in O(P log P)); this file serves every other size.
// This is the PARITY mode: O(P (H + W)) flops, not the throughput path (the 8x8 block kernel is).  At
// 64x64 the whole transform is 1 MFLOP per frame and runs on the vector ALUs (k_gemm_nt).  At native
// resolution (1080p = 12.4 GFLOP per frame pair) the two products are the one truly
// dense contraction of this code base, so planes of >= 128 x 128 go through k_gemm_nt_mfma: 128x128x16
// LDS-tiled blocks on v_mfma_f32_32x32x2_f32 — exact fp32 (a k-ordered fmaf chain), so the result keeps
// the 1e-4 parity bar, at the matrix pipe's rate instead of the vector ALUs'.
#include "vqa_dev.hpp"
#include "vqa_kernels.hpp"

namespace vqa {

enum { B_U8 = 0, B_U8_DIFF = 1, B_F32 = 2 };
enum { RED_STORE = 0, RED_SQ = 1, RED_ABS = 2 };

// out[M x N] = A[M x K] . B[N x K]^T ; 16x16 tile per 256-thread block.
// grid = (ceil(N/16), ceil(M/16), n_frames)
template <int BMODE, int RED>
__global__ __launch_bounds__(256) void k_gemm_nt(const float *__restrict__ A, int64_t a_frame_stride, int lda,
                                                 const void *__restrict__ Bv, const void *__restrict__ B2v,
                                                 int64_t b_frame_stride, int ldb, int M, int N, int K,
                                                 float *__restrict__ out, int64_t out_frame_stride, int ldo,
                                                 double *__restrict__ partials)
{
    __shared__ float As[16][17], Bs[16][17];
    __shared__ double red[4];
    const int f = blockIdx.z;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int m = blockIdx.y * 16 + ty, n = blockIdx.x * 16 + tx;
    const float *Af = A + (int64_t)f * a_frame_stride;
    float acc = 0.f;
    for (int k0 = 0; k0 < K; k0 += 16) {
        {
            const int am = blockIdx.y * 16 + ty, ak = k0 + tx;
            As[ty][tx] = (am < M && ak < K) ? Af[(int64_t)am * lda + ak] : 0.f;
            const int bn = blockIdx.x * 16 + ty, bk = k0 + tx;
            float bvv = 0.f;
            if (bn < N && bk < K) {
                if (BMODE == B_F32) {
                    bvv = ((const float *)Bv)[(int64_t)f * b_frame_stride + (int64_t)bn * ldb + bk];
                } else if (BMODE == B_U8) {
                    bvv = (float)((const uint8_t *)Bv)[(int64_t)f * b_frame_stride + (int64_t)bn * ldb + bk];
                } else {
                    bvv = (float)((const uint8_t *)B2v)[(int64_t)f * b_frame_stride + (int64_t)bn * ldb + bk] -
                          (float)((const uint8_t *)Bv)[(int64_t)f * b_frame_stride + (int64_t)bn * ldb + bk];
                }
            }
            Bs[ty][tx] = bvv;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; k++) acc = fmaf(As[ty][k], Bs[tx][k], acc);
        __syncthreads();
    }
    if (RED == RED_STORE) {
        if (m < M && n < N) out[(int64_t)f * out_frame_stride + (int64_t)m * ldo + n] = acc;
    } else {
        double v = 0;
        if (m < M && n < N) v = RED == RED_SQ ? (double)(acc * acc) : (double)fabsf(acc);
        const double t = block_sum(v, red);
        if (threadIdx.x == 0)
            partials[((int64_t)f * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = t;
    }
}

// ---------------------------------------------------------------------------
// out[M x N] = A[M x K] . B[N x K]^T on the matrix cores, fp32 in / fp32 accumulate.
// Block tile 128 x 128, K step 16; 4 waves as 2 x 2, each wave 64 x 64 = 2 x 2 MFMA tiles of 32 x 32
// (4 x 16 accumulator VGPRs).  Operand map of v_mfma_f32_32x32x2_f32: lane l holds A[i = l & 31][k = l >> 5]
// and B[k = l >> 5][j = l & 31]; C/D: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).
// LDS tiles are k-major [16][130]: the fragment reads are 32 consecutive floats per half-wave and the
// transposing stores (8 k-values of one row per thread) hit 32 distinct banks (pitch 130 = 2 mod 32).
// grid = (ceil(N/128), ceil(M/128), n_frames)
// ---------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int GB = 128, GK = 16, GP_ = 130;

template <int BMODE>
__device__ __forceinline__ void load8(const void *__restrict__ Bv, const void *__restrict__ B2v, int64_t base, int k,
                                      int K, bool row_ok, float v[8])
{
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = 0.f;
    if (!row_ok) return;
    if (BMODE == B_F32) {
        const float *p = (const float *)Bv + base + k;
        if (k + 8 <= K && (((uintptr_t)p) & 15) == 0) {
            const float4 a = ((const float4 *)p)[0], b = ((const float4 *)p)[1];
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++)
                if (k + j < K) v[j] = p[j];
        }
    } else {
        const uint8_t *p = (const uint8_t *)Bv + base + k;
        const uint8_t *q = BMODE == B_U8_DIFF ? (const uint8_t *)B2v + base + k : p;
        if (k + 8 <= K && ((((uintptr_t)p) | ((uintptr_t)q)) & 7) == 0) {
            const uint64_t pw = *(const uint64_t *)p, qw = BMODE == B_U8_DIFF ? *(const uint64_t *)q : 0ull;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int pv = (int)((pw >> (8 * j)) & 0xff), qv = (int)((qw >> (8 * j)) & 0xff);
                v[j] = BMODE == B_U8_DIFF ? (float)(qv - pv) : (float)pv;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++)
                if (k + j < K) v[j] = BMODE == B_U8_DIFF ? (float)((int)q[j] - (int)p[j]) : (float)p[j];
        }
    }
}

template <int BMODE, int RED>
__global__ __launch_bounds__(256) void k_gemm_nt_mfma(const float *__restrict__ A, int64_t a_frame_stride, int lda,
                                                      const void *__restrict__ Bv, const void *__restrict__ B2v,
                                                      int64_t b_frame_stride, int ldb, int M, int N, int K,
                                                      float *__restrict__ out, int64_t out_frame_stride, int ldo,
                                                      double *__restrict__ partials)
{
    __shared__ float As[GK * GP_], Bs[GK * GP_];
    __shared__ double red[4];
    const int f = blockIdx.z;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int wm = wv >> 1, wn = wv & 1;
    const int m0 = blockIdx.y * GB, n0 = blockIdx.x * GB;
    const int lrow = t >> 1, lk = (t & 1) * 8; // staging role: one tile row, 8 consecutive k
    const float *Af = A + (int64_t)f * a_frame_stride;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    const int64_t abase = (int64_t)(m0 + lrow) * lda;
    const int64_t bbase = (int64_t)f * b_frame_stride + (int64_t)(n0 + lrow) * ldb;
    const bool a_ok = m0 + lrow < M, b_ok = n0 + lrow < N;
    float av[8], bv[8];
    load8<B_F32>(Af, nullptr, abase, lk, K, a_ok, av);
    load8<BMODE>(Bv, B2v, bbase, lk, K, b_ok, bv);
    for (int k0 = 0; k0 < K; k0 += GK) {
        __syncthreads(); // previous step's fragment reads are done
#pragma unroll
        for (int j = 0; j < 8; j++) {
            As[(lk + j) * GP_ + lrow] = av[j];
            Bs[(lk + j) * GP_ + lrow] = bv[j];
        }
        __syncthreads();
        if (k0 + GK < K) { // next step's operands travel while this step's MFMAs run
            load8<B_F32>(Af, nullptr, abase, k0 + GK + lk, K, a_ok, av);
            load8<BMODE>(Bv, B2v, bbase, k0 + GK + lk, K, b_ok, bv);
        }
#pragma unroll
        for (int kk = 0; kk < GK; kk += 2) {
            const int kr = (kk + (lane >> 5)) * GP_ + (lane & 31);
            const float a0 = As[kr + wm * 64], a1 = As[kr + wm * 64 + 32];
            const float b0 = Bs[kr + wn * 64], b1 = Bs[kr + wn * 64 + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
    double sum = 0;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int n = n0 + wn * 64 + j * 32 + (lane & 31);
                if (m < M && n < N) {
                    const float v = acc[i][j][r];
                    if (RED == RED_STORE) out[(int64_t)f * out_frame_stride + (int64_t)m * ldo + n] = v;
                    else sum += RED == RED_SQ ? (double)(v * v) : (double)fabsf(v);
                }
            }
    if (RED != RED_STORE) {
        const double tsum = block_sum(sum, red);
        if (t == 0) partials[((int64_t)f * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = tsum;
    }
}

// one wave per frame: lane i adds partials i, i + 64, ... in double, then a fixed-order shuffle tree (bit-reproducible).
// (A thread per frame walked up to 640 partials serially: 0.14 ms per 64 frames behind the FFT passes' 3 ms.)
__global__ __launch_bounds__(64) void k_full_finalize(const double *__restrict__ pe, const double *__restrict__ pt, int tiles, int n,
                                                      vqa_frame_metrics *__restrict__ res, int write_energy, int write_temporal,
                                                      int first_has_prev)
{
    const int f = blockIdx.x;
    if (f >= n) return;
    double e = 0, t = 0;
    for (int i = threadIdx.x; i < tiles; i += 64) {
        if (write_energy) e += pe[(int64_t)f * tiles + i];
        if (write_temporal) t += pt[(int64_t)f * tiles + i];
    }
    e = wave_sum(e);
    t = wave_sum(t);
    if (threadIdx.x == 0) {
        if (write_energy) res[f].dct_energy = e;
        if (write_temporal) res[f].temporal_dct_l1 = (f > 0 || first_has_prev) ? t : 0.0;
    }
}

void launch_dct_full_finalize(hipStream_t st, const double *pe, const double *pt, int tiles, int n, vqa_frame_metrics *res,
                              bool energy, bool temporal, bool first_has_prev)
{
    hipLaunchKernelGGL(k_full_finalize, dim3(n), dim3(64), 0, st, pe, pt, tiles, n, res, (int)energy, (int)temporal,
                       (int)first_has_prev);
}

// planes: slot 0 = frame before the batch, slot i+1 = batch frame i (u8, pitch).
// cw: [w][w], ch: [h][h] orthonormal DCT matrices (device, float).
// scratch: n*w*h floats; pe/pt: n*tiles doubles each.
void launch_dct_full(hipStream_t st, const uint8_t *planes, int pitch, int64_t plane_stride, int n, int h, int w,
                     const float *cw, const float *ch, float *scratch, double *pe, double *pt, bool energy,
                     bool temporal, bool first_has_prev, vqa_frame_metrics *res)
{
    if (n <= 0 || (!energy && !temporal)) return;
    dim3 block(256);
    const bool mfma = h >= 128 && w >= 128;
    const int T = mfma ? GB : 16;
    dim3 g((h + T - 1) / T, (w + T - 1) / T, n); // both products: M = w, N = h  (K = w, then K = h)
    const int tiles = g.x * g.y;
    const uint8_t *curr = planes + plane_stride; // frame f -> slot f+1
    const uint8_t *prev = planes;                // frame f -> slot f
#define GEMM(KERNEL, BM, RD, ...) hipLaunchKernelGGL((KERNEL<BM, RD>), g, block, 0, st, __VA_ARGS__)
#define BOTH(BM, RD, ...)                                                                                             \
    do {                                                                                                              \
        if (mfma) GEMM(k_gemm_nt_mfma, BM, RD, __VA_ARGS__);                                                          \
        else GEMM(k_gemm_nt, BM, RD, __VA_ARGS__);                                                                    \
    } while (0)
    if (energy) {
        BOTH(B_U8, RED_STORE, cw, (int64_t)0, w, (const void *)curr, (const void *)nullptr, plane_stride, pitch, w, h, w,
             scratch, (int64_t)w * h, h, (double *)nullptr);
        BOTH(B_F32, RED_SQ, scratch, (int64_t)w * h, h, (const void *)ch, (const void *)nullptr, (int64_t)0, h, w, h, h,
             (float *)nullptr, (int64_t)0, 0, pe);
    }
    if (temporal) {
        // B = prev - curr  (Bv = curr, B2v = prev)
        BOTH(B_U8_DIFF, RED_STORE, cw, (int64_t)0, w, (const void *)curr, (const void *)prev, plane_stride, pitch, w, h, w,
             scratch, (int64_t)w * h, h, (double *)nullptr);
        BOTH(B_F32, RED_ABS, scratch, (int64_t)w * h, h, (const void *)ch, (const void *)nullptr, (int64_t)0, h, w, h, h,
             (float *)nullptr, (int64_t)0, 0, pt);
    }
#undef BOTH
#undef GEMM
    launch_dct_full_finalize(st, pe, pt, tiles, n, res, energy, temporal, first_has_prev);
}

} // namespace vqa
