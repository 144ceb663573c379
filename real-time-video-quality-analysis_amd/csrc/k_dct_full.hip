// k_dct_full.hip — full-frame orthonormal 2-D DCT-II, i.e. exactly what the
// reference's cv2.dct computes (complexity_metrics.py:363, :574-575), for the
// reference's own configuration (config.json: 64x64 after resize).
//
//   Y = C_h X C_w^T  as two NT products in fp32 on the vector ALUs:
//     Tt = C_w . X^T         (w x h)      X = gray, or prev - curr (linearity)
//     Yt = Tt . C_h^T        (w x h)      reduced on the fly: sum Y^2 or sum |Y|
//
// This is the PARITY mode for small planes; it is O(P (H + W)) flops and is not
// the throughput path (the 8x8 block kernel is).  At 64x64 the whole transform
// is 1 MFLOP per frame.
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

__global__ void k_full_finalize(const double *__restrict__ pe, const double *__restrict__ pt, int tiles, int n,
                                vqa_frame_metrics *__restrict__ res, int write_energy, int write_temporal,
                                int first_has_prev)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n) return;
    if (write_energy) {
        double e = 0;
        for (int i = 0; i < tiles; i++) e += pe[(int64_t)f * tiles + i];
        res[f].dct_energy = e;
    }
    if (write_temporal) {
        double t = 0;
        if (f > 0 || first_has_prev)
            for (int i = 0; i < tiles; i++) t += pt[(int64_t)f * tiles + i];
        res[f].temporal_dct_l1 = t;
    }
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
    dim3 g1((h + 15) / 16, (w + 15) / 16, n); // Tt: M = w, N = h, K = w
    dim3 g2((h + 15) / 16, (w + 15) / 16, n); // Yt: M = w, N = h, K = h
    const int tiles = g2.x * g2.y;
    const uint8_t *curr = planes + plane_stride; // frame f -> slot f+1
    const uint8_t *prev = planes;                // frame f -> slot f
    if (energy) {
        hipLaunchKernelGGL((k_gemm_nt<B_U8, RED_STORE>), g1, block, 0, st, cw, (int64_t)0, w, (const void *)curr,
                           (const void *)nullptr, plane_stride, pitch, w, h, w, scratch, (int64_t)w * h, h,
                           (double *)nullptr);
        hipLaunchKernelGGL((k_gemm_nt<B_F32, RED_SQ>), g2, block, 0, st, scratch, (int64_t)w * h, h, (const void *)ch,
                           (const void *)nullptr, (int64_t)0, h, w, h, h, (float *)nullptr, (int64_t)0, 0, pe);
    }
    if (temporal) {
        // B = prev - curr  (Bv = curr, B2v = prev)
        hipLaunchKernelGGL((k_gemm_nt<B_U8_DIFF, RED_STORE>), g1, block, 0, st, cw, (int64_t)0, w, (const void *)curr,
                           (const void *)prev, plane_stride, pitch, w, h, w, scratch, (int64_t)w * h, h,
                           (double *)nullptr);
        hipLaunchKernelGGL((k_gemm_nt<B_F32, RED_ABS>), g2, block, 0, st, scratch, (int64_t)w * h, h, (const void *)ch,
                           (const void *)nullptr, (int64_t)0, h, w, h, h, (float *)nullptr, (int64_t)0, 0, pt);
    }
    hipLaunchKernelGGL(k_full_finalize, dim3((n + 63) / 64), dim3(64), 0, st, pe, pt, tiles, n, res, (int)energy,
                       (int)temporal, (int)first_has_prev);
}

} // namespace vqa
