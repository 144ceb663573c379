// k_gray_hist.hip — BGR -> gray conversion, cv2.resize, and LDS-privatised
// 256-bin histograms for gfx950.
//
// Reference functions replaced (complexity_metrics.py):
//   :358-359  cvtColor(BGR2GRAY) -> resize            (DCT / temporal-DCT input plane "A")
//   :404-405  resize -> cvtColor(BGR2GRAY) -> calcHist (gray histogram, plane "B")
//   :430,455-457 resize -> 3x calcHist                 (colour histograms)
//   :490-493  resize -> cvtColor                       (Canny input, plane "B")
//   :327-328  cvtColor at full resolution              (motion input)
//
// Roofline: HBM. Native-resolution kernel reads 3P (packed BGR) and writes P
// (gray) per frame, 16 pixels = 48 B in / 16 B out per lane per iteration with
// dwordx4 accesses; bins live in per-wave LDS copies and reach HBM once per
// block as integer atomics.
#include "vqa_dev.hpp"
#include "vqa_kernels.hpp"
#include "vqa_math.hpp"

namespace vqa {

// flags
enum { F_GRAY_HIST = 1, F_COLOR_HIST = 2, F_SUM2 = 4 };

__device__ __forceinline__ uint32_t byte_of(const uint32_t *d, int k) { return (d[k >> 2] >> ((k & 3) * 8)) & 0xffu; }

// grid = (blocks_per_frame, n_frames), block = 256.
template <bool VEC>
__global__ __launch_bounds__(256) void k_bgr2gray_hist(const uint8_t *__restrict__ bgr, int h, int w,
                                                       int64_t frame_stride, int64_t row_stride,
                                                       uint8_t *__restrict__ gray, int gp, int64_t plane_stride,
                                                       vqa_frame_metrics *__restrict__ res, uint32_t flags)
{
    // [gray,b,g,r][bin][copy]: 8 copies of every bin, chosen by lane & 7 and INTERLEAVED so that the copies
    // of one bin sit in 8 different LDS banks.  On smooth content most lanes of a wave hit the same few
    // bins; this turns a 64-way same-address serialisation into at most 8-way.
    __shared__ uint32_t lh[4 * 256 * 8];
    __shared__ unsigned long long red[4];
    const int f = blockIdx.y;
    const bool hg = (flags & F_GRAY_HIST) && res, hc = (flags & F_COLOR_HIST) && res;
    const bool any_hist = hg || hc;
    if (any_hist) {
        for (int i = threadIdx.x; i < 4 * 256 * 8; i += 256) lh[i] = 0;
        __syncthreads();
    }
    const unsigned cp = threadIdx.x & 7u;
    const uint8_t *src = bgr + (int64_t)f * frame_stride;
    uint8_t *dst = gray + (int64_t)f * plane_stride;
    const int cpr = (w + 15) >> 4;
    const int total = h * cpr;
    unsigned long long s2 = 0;
    for (int c = blockIdx.x * 256 + threadIdx.x; c < total; c += gridDim.x * 256) {
        const int y = c / cpr;
        const int x0 = (c - y * cpr) << 4;
        const uint8_t *p = src + (int64_t)y * row_stride + (int64_t)x0 * 3;
        uint32_t d[12];
        int npx = 16;
        if (VEC) {
            const uint4 v0 = ((const uint4 *)p)[0], v1 = ((const uint4 *)p)[1], v2 = ((const uint4 *)p)[2];
            d[0] = v0.x; d[1] = v0.y; d[2] = v0.z; d[3] = v0.w;
            d[4] = v1.x; d[5] = v1.y; d[6] = v1.z; d[7] = v1.w;
            d[8] = v2.x; d[9] = v2.y; d[10] = v2.z; d[11] = v2.w;
        } else {
            npx = min(16, w - x0);
#pragma unroll
            for (int k = 0; k < 12; k++) d[k] = 0;
            for (int k = 0; k < npx * 3; k++) d[k >> 2] |= (uint32_t)p[k] << ((k & 3) * 8);
        }
        uint32_t g[16];
        uint32_t sq = 0;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const uint32_t b = byte_of(d, 3 * i), gg = byte_of(d, 3 * i + 1), r = byte_of(d, 3 * i + 2);
            g[i] = bgr2gray_u8(b, gg, r);
            if (VEC || i < npx) {
                sq += g[i] * g[i];
                if (hg) atomicAdd(&lh[(g[i] << 3) | cp], 1u);
                if (hc) {
                    atomicAdd(&lh[((256u + b) << 3) | cp], 1u);
                    atomicAdd(&lh[((512u + gg) << 3) | cp], 1u);
                    atomicAdd(&lh[((768u + r) << 3) | cp], 1u);
                }
            }
        }
        s2 += sq;
        uint8_t *q = dst + (int64_t)y * gp + x0;
        if (VEC) {
            uint4 o;
            o.x = g[0] | (g[1] << 8) | (g[2] << 16) | (g[3] << 24);
            o.y = g[4] | (g[5] << 8) | (g[6] << 16) | (g[7] << 24);
            o.z = g[8] | (g[9] << 8) | (g[10] << 16) | (g[11] << 24);
            o.w = g[12] | (g[13] << 8) | (g[14] << 16) | (g[15] << 24);
            *(uint4 *)q = o;
        } else {
            for (int i = 0; i < npx; i++) q[i] = (uint8_t)g[i];
        }
    }
    if (!res) return;
    if (any_hist) {
        __syncthreads();
        for (int i = threadIdx.x; i < 4 * 256; i += 256) {
            const uint4 c0 = *(const uint4 *)&lh[i * 8], c1 = *(const uint4 *)&lh[i * 8 + 4];
            const uint32_t v = c0.x + c0.y + c0.z + c0.w + c1.x + c1.y + c1.z + c1.w;
            if (v) {
                const int which = i >> 8, bin = i & 255;
                uint32_t *gdst = which == 0 ? &res[f].hist_gray[bin] : &res[f].hist_bgr[which - 1][bin];
                atomicAdd(gdst, v);
            }
        }
    }
    if (flags & F_SUM2) {
        const unsigned long long t = block_sum_u64(s2, red);
        if (threadIdx.x == 0 && t) atomicAdd((unsigned long long *)&res[f].sum_gray2, t);
    }
}

// ---------------------------------------------------------------------------
// cv2.resize path (config.json default 64x64).  One thread per output pixel;
// gathers the 2x2 source neighbourhood straight from the packed BGR frame.
//   plane A = resize(gray(frame))  (:358-359)   plane B = gray(resize(frame)) (:404-405,:490-493)
// Tables (xofs, xa, yofs, yb) are built on the host exactly as OpenCV builds them.
// mode: 0 = fixed-point bilinear, 1 = exact 2x decimation (INTER_AREA fast path)
// grid = (ceil(rw*rh/256), n_frames)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_resize_planes(const uint8_t *__restrict__ bgr, int h, int w,
                                                       int64_t frame_stride, int64_t row_stride, int rw, int rh,
                                                       const int32_t *__restrict__ xofs, const int32_t *__restrict__ xa,
                                                       const int32_t *__restrict__ yofs, const int32_t *__restrict__ yb,
                                                       int mode, uint8_t *__restrict__ planeA,
                                                       uint8_t *__restrict__ planeB, int pp, int64_t plane_stride,
                                                       vqa_frame_metrics *__restrict__ res, uint32_t flags)
{
    __shared__ uint32_t lh[4][256];
    __shared__ unsigned long long red[4];
    const int f = blockIdx.y;
    const bool hg = (flags & F_GRAY_HIST) && res, hc = (flags & F_COLOR_HIST) && res;
    if (hg || hc) {
        for (int i = threadIdx.x; i < 4 * 256; i += 256) (&lh[0][0])[i] = 0;
        __syncthreads();
    }
    const int idx = blockIdx.x * 256 + threadIdx.x;
    unsigned long long s2 = 0;
    if (idx < rw * rh) {
        const int oy = idx / rw, ox = idx - oy * rw;
        int sx0, sx1, sy0, sy1, a0, a1, b0, b1;
        if (mode == 1) {
            sx0 = 2 * ox; sx1 = sx0 + 1; sy0 = 2 * oy; sy1 = sy0 + 1;
            a0 = a1 = b0 = b1 = 0;
        } else {
            sx0 = xofs[ox]; sx1 = min(sx0 + 1, w - 1);
            a0 = xa[2 * ox]; a1 = xa[2 * ox + 1];
            const int sy = yofs[oy];
            sy0 = min(max(sy, 0), h - 1); sy1 = min(max(sy + 1, 0), h - 1);
            b0 = yb[2 * oy]; b1 = yb[2 * oy + 1];
        }
        const uint8_t *src = bgr + (int64_t)f * frame_stride;
        const uint8_t *p00 = src + (int64_t)sy0 * row_stride + (int64_t)sx0 * 3;
        const uint8_t *p01 = src + (int64_t)sy0 * row_stride + (int64_t)sx1 * 3;
        const uint8_t *p10 = src + (int64_t)sy1 * row_stride + (int64_t)sx0 * 3;
        const uint8_t *p11 = src + (int64_t)sy1 * row_stride + (int64_t)sx1 * 3;
        uint32_t c00[3], c01[3], c10[3], c11[3];
#pragma unroll
        for (int k = 0; k < 3; k++) { c00[k] = p00[k]; c01[k] = p01[k]; c10[k] = p10[k]; c11[k] = p11[k]; }
        const uint32_t g00 = bgr2gray_u8(c00[0], c00[1], c00[2]), g01 = bgr2gray_u8(c01[0], c01[1], c01[2]);
        const uint32_t g10 = bgr2gray_u8(c10[0], c10[1], c10[2]), g11 = bgr2gray_u8(c11[0], c11[1], c11[2]);
        uint32_t ga, rc[3];
        if (mode == 1) {
            ga = (g00 + g01 + g10 + g11 + 2) >> 2;
#pragma unroll
            for (int k = 0; k < 3; k++) rc[k] = (c00[k] + c01[k] + c10[k] + c11[k] + 2) >> 2;
        } else {
            ga = resize_vcombine((int)(g00 * a0 + g01 * a1), (int)(g10 * a0 + g11 * a1), b0, b1);
#pragma unroll
            for (int k = 0; k < 3; k++)
                rc[k] = resize_vcombine((int)(c00[k] * a0 + c01[k] * a1), (int)(c10[k] * a0 + c11[k] * a1), b0, b1);
        }
        const uint32_t gb = bgr2gray_u8(rc[0], rc[1], rc[2]);
        planeA[(int64_t)f * plane_stride + (int64_t)oy * pp + ox] = (uint8_t)ga;
        if (planeB) planeB[(int64_t)f * plane_stride + (int64_t)oy * pp + ox] = (uint8_t)gb;
        s2 = (unsigned long long)(ga * ga);
        if (hg) atomicAdd(&lh[0][gb], 1u);
        if (hc) { atomicAdd(&lh[1][rc[0]], 1u); atomicAdd(&lh[2][rc[1]], 1u); atomicAdd(&lh[3][rc[2]], 1u); }
    }
    if (!res) return;
    if (hg || hc) {
        __syncthreads();
        for (int i = threadIdx.x; i < 4 * 256; i += 256) {
            const uint32_t v = (&lh[0][0])[i];
            if (v) {
                const int which = i >> 8, bin = i & 255;
                atomicAdd(which == 0 ? &res[f].hist_gray[bin] : &res[f].hist_bgr[which - 1][bin], v);
            }
        }
    }
    if (flags & F_SUM2) {
        const unsigned long long t = block_sum_u64(s2, red);
        if (threadIdx.x == 0 && t) atomicAdd((unsigned long long *)&res[f].sum_gray2, t);
    }
}

// Histogram of an already-built gray plane is never needed separately: the two
// kernels above produce every histogram the reference asks for.

void launch_bgr2gray_hist(hipStream_t st, const uint8_t *bgr, int n, int h, int w, int64_t frame_stride,
                          int64_t row_stride, uint8_t *gray, int gp, int64_t plane_stride, vqa_frame_metrics *res,
                          bool gray_hist, bool color_hist, bool sum2)
{
    if (n <= 0) return;
    const uint32_t flags = (gray_hist ? F_GRAY_HIST : 0) | (color_hist ? F_COLOR_HIST : 0) | (sum2 ? F_SUM2 : 0);
    const bool vec = (w % 16 == 0) && (((uintptr_t)bgr | (uintptr_t)gray) % 16 == 0) && (frame_stride % 16 == 0) &&
                     (row_stride % 16 == 0) && (gp % 16 == 0) && (plane_stride % 16 == 0);
    const int chunks = h * ((w + 15) / 16);
    // ~8 chunks (128 px) per thread; at least one block per frame
    int bpf = (chunks + 256 * 8 - 1) / (256 * 8);
    bpf = bpf < 1 ? 1 : (bpf > 256 ? 256 : bpf);
    dim3 grid(bpf, n), block(256);
    if (vec)
        hipLaunchKernelGGL(k_bgr2gray_hist<true>, grid, block, 0, st, bgr, h, w, frame_stride, row_stride, gray, gp,
                           plane_stride, res, flags);
    else
        hipLaunchKernelGGL(k_bgr2gray_hist<false>, grid, block, 0, st, bgr, h, w, frame_stride, row_stride, gray, gp,
                           plane_stride, res, flags);
}

void launch_resize_planes(hipStream_t st, const uint8_t *bgr, int n, int h, int w, int64_t frame_stride,
                          int64_t row_stride, int rw, int rh, const int32_t *xofs, const int32_t *xa,
                          const int32_t *yofs, const int32_t *yb, int mode, uint8_t *planeA, uint8_t *planeB, int pp,
                          int64_t plane_stride, vqa_frame_metrics *res, bool gray_hist, bool color_hist, bool sum2)
{
    if (n <= 0) return;
    const uint32_t flags = (gray_hist ? F_GRAY_HIST : 0) | (color_hist ? F_COLOR_HIST : 0) | (sum2 ? F_SUM2 : 0);
    dim3 grid((rw * rh + 255) / 256, n), block(256);
    hipLaunchKernelGGL(k_resize_planes, grid, block, 0, st, bgr, h, w, frame_stride, row_stride, rw, rh, xofs, xa, yofs,
                       yb, mode, planeA, planeB, pp, plane_stride, res, flags);
}

} // namespace vqa
