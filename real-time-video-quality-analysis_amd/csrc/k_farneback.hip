// k_farneback.hip — Farneback dense optical flow (reference-true motion) for gfx950.
//
// Reference function replaced: process_frame_complexity, complexity_metrics.py:340-343
//   flow = cv2.calcOpticalFlowFarneback(prev_gray, curr_gray, None, 0.5, 3, 15, 3, 5, 1.2, 0)
//   np.mean(cv2.cartToPolar(flow[...,0], flow[...,1])[0])
// BASELINE.json's north_star replaces this metric by block-SAD motion (k_sad.hip, the default);
// this file is SURVEY.md §8(f) row N4: the reference's own motion metric, selectable with
// vqa_params.motion_mode = VQA_MOTION_FARNEBACK.  The algorithm (OpenCV 4.x video/optflowgf.cpp, CPU path)
// is spelled out next to its CPU restatement, oracle/vqa_oracle.c vqo_farneback_mean_mag.
//
// Every kernel evaluates the same float / double expressions in the same order as that restatement, and
// this translation unit is compiled with floating-point contraction OFF, so the float planes (blurred
// images, expansions, products) are bit-identical to the oracle's: the 2x2 solve is ill-conditioned wherever
// the image is flat or has a single orientation, and a fused multiply-add in the products would show up
// in the flow there.  (The double box sums differ from the oracle's tap order at the 1e-16 level only.)
//
// Structure per pyramid level (coarse to fine), for a chunk of frame pairs whose planes stay resident:
//   k_fb_blur_h / k_fb_blur_v : float(gray) -> separable Gaussian, BORDER_REFLECT_101   (per plane, shared
//   k_fb_resize               : INTER_LINEAR float resize (2x2 mean when exactly halving)  by both pairs
//   k_fb_polyexp              : 11x11 polynomial expansion -> 5 coefficients per pixel      it belongs to)
//   (the coarser level's flow is upsampled and doubled inside the level's first k_fb_update)
//   k_fb_update               : bilinear warp of the second expansion by the flow -> 5 products per pixel
//   k_fb_blur_solve           : 15x15 box sums (double, replicated border) + regularised 2x2 solve -> flow
// and k_fb_mag for the mean magnitude.  All of it is stencil work on fp32 planes: HBM / LDS / VALU, no MFMA.
#pragma clang fp contract(off)
#include "vqa_dev.hpp"
#include "vqa_kernels.hpp"

namespace vqa {

__device__ __forceinline__ int fb_reflect101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
    return i;
}

// ---- Gaussian blur, horizontal pass straight from the u8 gray plane -------------------------------
// Only the columns in `cols` (all when null) are produced: at the coarse levels the bilinear resize that
// follows samples 2 of every 4 or 8 columns and rows, and nothing else reads the blurred plane.
// grid = (ceil(nc/256), h, planes)
__global__ __launch_bounds__(256) void k_fb_blur_h(const uint8_t *__restrict__ gray, int pitch, int64_t plane_stride,
                                                   int h, int w, fb_taps T, const int32_t *__restrict__ cols, int nc,
                                                   float *__restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (i >= nc) return;
    const int x = cols ? cols[i] : i;
    const uint8_t *row = gray + (int64_t)blockIdx.z * plane_stride + (int64_t)y * pitch;
    const int r = T.ksize >> 1;
    float a = (float)row[x] * T.k[r];
    for (int k = 1; k <= r; k++)
        a += ((float)row[fb_reflect101(x - k, w)] + (float)row[fb_reflect101(x + k, w)]) * T.k[r + k];
    out[((int64_t)blockIdx.z * h + y) * w + x] = a;
}

// grid = (ceil(nc/256), nr, planes)
__global__ __launch_bounds__(256) void k_fb_blur_v(const float *__restrict__ in, int h, int w, fb_taps T,
                                                   const int32_t *__restrict__ cols, int nc,
                                                   const int32_t *__restrict__ rows, float *__restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nc) return;
    const int x = cols ? cols[i] : i, y = rows ? rows[blockIdx.y] : (int)blockIdx.y;
    const float *img = in + (int64_t)blockIdx.z * h * w;
    const int r = T.ksize >> 1;
    float a = img[(int64_t)y * w + x] * T.k[r];
    for (int k = 1; k <= r; k++)
        a += (img[(int64_t)fb_reflect101(y - k, h) * w + x] + img[(int64_t)fb_reflect101(y + k, h) * w + x]) * T.k[r + k];
    out[((int64_t)blockIdx.z * h + y) * w + x] = a;
}

// ---- cv2.resize INTER_LINEAR on float data with CN interleaved channels; result times `mul` --------
// mode 0: bilinear with host-built tables; mode 1: exact 2x decimation (INTER_AREA fast path)
// grid = (ceil(dw/256), dh, images)
template <int CN>
__global__ __launch_bounds__(256) void k_fb_resize(const float *__restrict__ src, int sh, int sw, float *__restrict__ dst,
                                                   int dh, int dw, const int32_t *__restrict__ xofs,
                                                   const float *__restrict__ xa, const int32_t *__restrict__ yofs,
                                                   const float *__restrict__ yb, int mode, float mul, int apply_mul)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= dw) return;
    const float *s = src + (int64_t)blockIdx.z * sh * sw * CN;
    float *d = dst + (((int64_t)blockIdx.z * dh + y) * dw + x) * CN;
    if (mode == 1) {
#pragma unroll
        for (int c = 0; c < CN; c++) {
            const float *p = s + ((int64_t)(2 * y) * sw + 2 * x) * CN + c;
            float v = (p[0] + p[CN] + p[(int64_t)sw * CN] + p[(int64_t)sw * CN + CN]) * 0.25f;
            if (apply_mul) v *= mul;
            d[c] = v;
        }
        return;
    }
    const int x0 = xofs[x], x1 = min(x0 + 1, sw - 1);
    const int y0 = min(max(yofs[y], 0), sh - 1), y1 = min(max(yofs[y] + 1, 0), sh - 1);
    const float a0 = xa[2 * x], a1 = xa[2 * x + 1], b0 = yb[2 * y], b1 = yb[2 * y + 1];
#pragma unroll
    for (int c = 0; c < CN; c++) {
        const float r0 = s[((int64_t)y0 * sw + x0) * CN + c] * a0 + s[((int64_t)y0 * sw + x1) * CN + c] * a1;
        const float r1 = s[((int64_t)y1 * sw + x0) * CN + c] * a0 + s[((int64_t)y1 * sw + x1) * CN + c] * a1;
        float v = r0 * b0 + r1 * b1;
        if (apply_mul) v *= mul;
        d[c] = v;
    }
}

// ---- polynomial expansion (FarnebackPolyExp, n = 5) ------------------------------------------------
// Tile of PE_TY rows x PE_TX columns per workgroup: the vertical pass (float) fills LDS for the tile's
// columns plus a 5-column replicated halo, the horizontal pass (double accumulators) reads it back.
constexpr int PE_TX = 64, PE_TY = 4, PE_N = 5;

// grid = (ceil(w/PE_TX), ceil(h/PE_TY), planes), block = 256 (thread = one output pixel)
__global__ __launch_bounds__(256) void k_fb_polyexp(const float *__restrict__ in, int h, int w, fb_poly C,
                                                    float *__restrict__ out)
{
    __shared__ float row[PE_TY][PE_TX + 2 * PE_N][3];
    const float *img = in + (int64_t)blockIdx.z * h * w;
    const int x0 = blockIdx.x * PE_TX, y0 = blockIdx.y * PE_TY;
    const float *g = C.g + PE_N, *xg = C.xg + PE_N, *xxg = C.xxg + PE_N;
    for (int i = threadIdx.x; i < PE_TY * (PE_TX + 2 * PE_N); i += 256) {
        const int ty = i / (PE_TX + 2 * PE_N), tx = i - ty * (PE_TX + 2 * PE_N);
        const int y = y0 + ty;
        const int x = min(max(x0 + tx - PE_N, 0), w - 1); // the horizontal pass replicates the border triples
        float t0 = 0.f, t1 = 0.f, t2 = 0.f;
        if (y < h) {
            t0 = img[(int64_t)y * w + x] * g[0];
#pragma unroll
            for (int k = 1; k <= PE_N; k++) {
                const float a = img[(int64_t)max(y - k, 0) * w + x], b = img[(int64_t)min(y + k, h - 1) * w + x];
                const float p = a + b;
                t0 = t0 + g[k] * p;
                t1 = t1 + xg[k] * (b - a);
                t2 = t2 + xxg[k] * p;
            }
        }
        row[ty][tx][0] = t0; row[ty][tx][1] = t1; row[ty][tx][2] = t2;
    }
    __syncthreads();
    const int ty = threadIdx.x / PE_TX, tx = threadIdx.x % PE_TX;
    const int x = x0 + tx, y = y0 + ty;
    if (x >= w || y >= h) return;
    const float(*r)[3] = &row[ty][tx + PE_N];
    double b1 = r[0][0] * g[0], b2 = 0, b3 = r[0][1] * g[0], b4 = 0, b5 = r[0][2] * g[0], b6 = 0;
#pragma unroll
    for (int k = 1; k <= PE_N; k++) {
        const double tg = r[k][0] + r[-k][0];
        b1 += tg * g[k];
        b4 += tg * xxg[k];
        b2 += (r[k][0] - r[-k][0]) * xg[k];
        b3 += (r[k][1] + r[-k][1]) * g[k];
        b6 += (r[k][1] - r[-k][1]) * xg[k];
        b5 += (r[k][2] + r[-k][2]) * g[k];
    }
    float *d = out + (((int64_t)blockIdx.z * h + y) * w + x) * 5;
    d[1] = (float)(b2 * C.ig11);
    d[0] = (float)(b3 * C.ig11);
    d[3] = (float)(b1 * C.ig03 + b4 * C.ig33);
    d[2] = (float)(b1 * C.ig03 + b5 * C.ig33);
    d[4] = (float)(b6 * C.ig55);
}

// ---- FarnebackUpdateMatrices: pair p uses expansions of planes p and p + 1 -------------------------
// grid = (ceil(w/256), h, pairs)
// SRC: where the flow comes from.  0 = the level's flow field; 1 = the coarser level's flow, resized
// (INTER_LINEAR, the k_fb_resize<2> arithmetic) and doubled on the fly — the first rebuild of a level is its
// only reader, so the upsampled field is never written; 2 = zero (coarsest level).
template <int SRC>
__global__ __launch_bounds__(256) void k_fb_update(const float *__restrict__ R, const float *__restrict__ flow, int h,
                                                   int w, float *__restrict__ M, int ch, int cw,
                                                   const int32_t *__restrict__ xofs, const float *__restrict__ xa,
                                                   const int32_t *__restrict__ yofs, const float *__restrict__ yb,
                                                   float mul)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const int64_t P = (int64_t)h * w;
    const float *R0 = R + (int64_t)blockIdx.z * P * 5, *R1 = R0 + P * 5;
    const int64_t pix = (int64_t)blockIdx.z * P + (int64_t)y * w + x;
    const float *r0 = R0 + ((int64_t)y * w + x) * 5;
    float dx = 0.f, dy = 0.f;
    if (SRC == 0) {
        dx = flow[pix * 2];
        dy = flow[pix * 2 + 1];
    } else if (SRC == 1) {
        const float *s = flow + (int64_t)blockIdx.z * ch * cw * 2;
        const int x0 = xofs[x], x1c = min(x0 + 1, cw - 1);
        const int y0 = min(max(yofs[y], 0), ch - 1), y1c = min(max(yofs[y] + 1, 0), ch - 1);
        const float a0 = xa[2 * x], a1 = xa[2 * x + 1], b0 = yb[2 * y], b1 = yb[2 * y + 1];
        float v[2];
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const float q0 = s[((int64_t)y0 * cw + x0) * 2 + c] * a0 + s[((int64_t)y0 * cw + x1c) * 2 + c] * a1;
            const float q1 = s[((int64_t)y1c * cw + x0) * 2 + c] * a0 + s[((int64_t)y1c * cw + x1c) * 2 + c] * a1;
            v[c] = (q0 * b0 + q1 * b1) * mul;
        }
        dx = v[0];
        dy = v[1];
    }
    float fx = x + dx, fy = y + dy;
    const int x1 = (int)floorf(fx), y1 = (int)floorf(fy);
    float r2, r3, r4, r5, r6;
    fx -= x1; fy -= y1;
    if ((unsigned)x1 < (unsigned)(w - 1) && (unsigned)y1 < (unsigned)(h - 1)) {
        const int64_t step1 = (int64_t)w * 5;
        const float *p = R1 + (int64_t)y1 * step1 + (int64_t)x1 * 5;
        const float a00 = (1.f - fx) * (1.f - fy), a01 = fx * (1.f - fy), a10 = (1.f - fx) * fy, a11 = fx * fy;
        r2 = a00 * p[0] + a01 * p[5] + a10 * p[step1] + a11 * p[step1 + 5];
        r3 = a00 * p[1] + a01 * p[6] + a10 * p[step1 + 1] + a11 * p[step1 + 6];
        r4 = a00 * p[2] + a01 * p[7] + a10 * p[step1 + 2] + a11 * p[step1 + 7];
        r5 = a00 * p[3] + a01 * p[8] + a10 * p[step1 + 3] + a11 * p[step1 + 8];
        r6 = a00 * p[4] + a01 * p[9] + a10 * p[step1 + 4] + a11 * p[step1 + 9];
        r4 = (r0[2] + r4) * 0.5f;
        r5 = (r0[3] + r5) * 0.5f;
        r6 = (r0[4] + r6) * 0.25f;
    } else {
        r2 = r3 = 0.f;
        r4 = r0[2]; r5 = r0[3]; r6 = r0[4] * 0.5f;
    }
    r2 = (r0[0] - r2) * 0.5f;
    r3 = (r0[1] - r3) * 0.5f;
    r2 += r4 * dy + r6 * dx;
    r3 += r6 * dy + r5 * dx;
    if ((unsigned)(x - 5) >= (unsigned)(w - 10) || (unsigned)(y - 5) >= (unsigned)(h - 10)) {
        const float border[5] = {0.14f, 0.14f, 0.4472f, 0.4472f, 0.4472f};
        const float scale = (x < 5 ? border[x] : 1.f) * (x >= w - 5 ? border[w - x - 1] : 1.f) *
                            (y < 5 ? border[y] : 1.f) * (y >= h - 5 ? border[h - y - 1] : 1.f);
        r2 *= scale; r3 *= scale; r4 *= scale; r5 *= scale; r6 *= scale;
    }
    float *m = M + pix * 5;
    m[0] = r4 * r4 + r6 * r6;
    m[1] = (r4 + r5) * r6;
    m[2] = r5 * r5 + r6 * r6;
    m[3] = r4 * r2 + r6 * r3;
    m[4] = r6 * r2 + r5 * r3;
}

// ---- FarnebackUpdateFlow_Blur: 15x15 box sums of the products + 2x2 solve --------------------------
// Tile BS_TY x BS_TX outputs per workgroup.  The tile's products (+-7 halo, rows and columns clamped =
// replicated border) are staged in LDS once; stage 1 slides a 15-wide window along every (row, channel)
// in double, stage 2 slides a 15-high window down every (column, channel), stage 3 scales and solves.
// Sums are carried in double as in OpenCV; the sliding order differs from the oracle's tap order only at
// the 1e-16 level, far below what the regularised solve can amplify.
constexpr int BS_TX = 32, BS_TY = 16, BS_M = 7;
constexpr int BS_IW = BS_TX + 2 * BS_M, BS_IH = BS_TY + 2 * BS_M;

constexpr int BS_NT = 1024; // 16 waves share one 66 KB tile: the global->LDS fill and the solve dominate, both want threads
// grid = (ceil(w/BS_TX), ceil(h/BS_TY), pairs), block = BS_NT
__global__ __launch_bounds__(BS_NT) void k_fb_blur_solve(const float *__restrict__ M, int h, int w, float *__restrict__ flow)
{
    __shared__ float tile[BS_IH][BS_IW][5];   // 27.6 KB
    __shared__ double hs[BS_IH][BS_TX][5];    // 38.4 KB: horizontal sums, then (in place) the window sums
    const int64_t P = (int64_t)h * w;
    const float *Mp = M + (int64_t)blockIdx.z * P * 5;
    // Workgroups are dealt round-robin over the 8 XCDs (each with its own L2) in launch order.  Neighbouring
    // tiles share 7-pixel halos, so every XCD gets a contiguous band of the tile raster instead of every
    // eighth tile: halo rows are re-read from that XCD's L2 (HBM fetch of this kernel 261 -> 126 B per
    // pixel-pair; its run time is bound by the sliding-sum chains, not by that traffic, and did not move).
    int bx = blockIdx.x, by = blockIdx.y;
    {
        const unsigned nt = gridDim.x * gridDim.y, lin = blockIdx.x + gridDim.x * blockIdx.y;
        if ((nt & 7u) == 0) {
            const unsigned t = (lin & 7u) * (nt >> 3) + (lin >> 3);
            bx = (int)(t % gridDim.x);
            by = (int)(t / gridDim.x);
        }
    }
    const int x0 = bx * BS_TX, y0 = by * BS_TY;
    for (int i = threadIdx.x; i < BS_IH * BS_IW; i += BS_NT) {
        const int ty = i / BS_IW, tx = i - ty * BS_IW;
        const int yy = min(max(y0 + ty - BS_M, 0), h - 1), xx = min(max(x0 + tx - BS_M, 0), w - 1);
        const float *m = Mp + ((int64_t)yy * w + xx) * 5;
#pragma unroll
        for (int c = 0; c < 5; c++) tile[ty][tx][c] = m[c];
    }
    __syncthreads();
    // stage 1: thread = (row, channel); hs[row][x][c] = sum of tile[row][x .. x+14][c]
    for (int i = threadIdx.x; i < BS_IH * 5; i += BS_NT) {
        const int ty = i / 5, c = i - ty * 5;
        double a = 0;
#pragma unroll
        for (int k = 0; k < 2 * BS_M + 1; k++) a += tile[ty][k][c];
        hs[ty][0][c] = a;
        for (int x = 1; x < BS_TX; x++) {
            a += (double)tile[ty][x + 2 * BS_M][c] - (double)tile[ty][x - 1][c];
            hs[ty][x][c] = a;
        }
    }
    __syncthreads();
    // stage 2: thread = (column, channel); the window sum of rows y .. y+14 overwrites hs[y]: rows are
    // consumed top-down and hs[y] is read for the last time when output y is produced
    for (int i = threadIdx.x; i < BS_TX * 5; i += BS_NT) {
        const int tx = i / 5, c = i - tx * 5;
        double a = 0;
#pragma unroll
        for (int k = 0; k < 2 * BS_M + 1; k++) a += hs[k][tx][c];
        for (int y = 0; y < BS_TY; y++) {
            const double top = hs[y][tx][c];
            hs[y][tx][c] = a;
            if (y + 1 < BS_TY) a += hs[y + 2 * BS_M + 1][tx][c] - top;
        }
    }
    __syncthreads();
    const double scale = 1. / 225.;
    for (int i = threadIdx.x; i < BS_TY * BS_TX; i += BS_NT) {
        const int ty = i / BS_TX, tx = i - ty * BS_TX;
        const int x = x0 + tx, y = y0 + ty;
        if (x >= w || y >= h) continue;
        double v[5];
#pragma unroll
        for (int c = 0; c < 5; c++) v[c] = hs[ty][tx][c] * scale;
        const double idet = 1. / (v[0] * v[2] - v[1] * v[1] + 1e-3);
        float *f = flow + ((int64_t)blockIdx.z * P + (int64_t)y * w + x) * 2;
        f[0] = (float)((v[0] * v[4] - v[1] * v[3]) * idet);
        f[1] = (float)((v[2] * v[3] - v[1] * v[4]) * idet);
    }
}

// ---- mean magnitude --------------------------------------------------------------------------------
// grid = (FB_MAG_BLOCKS, pairs); partials[pair][block] then a fixed-order finalize (bit-reproducible)
constexpr int FB_MAG_BLOCKS = 64;

__global__ __launch_bounds__(256) void k_fb_mag(const float *__restrict__ flow, int64_t P, double *__restrict__ partials)
{
    __shared__ double red[4];
    const float *f = flow + (int64_t)blockIdx.y * P * 2;
    double s = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < P; i += (int64_t)FB_MAG_BLOCKS * 256)
        s += (double)sqrtf(f[2 * i] * f[2 * i] + f[2 * i + 1] * f[2 * i + 1]);
    const double t = block_sum(s, red);
    if (threadIdx.x == 0) partials[(int64_t)blockIdx.y * FB_MAG_BLOCKS + blockIdx.x] = t;
}

__global__ void k_fb_mag_finalize(const double *__restrict__ partials, int pairs, double inv_count, int first_valid,
                                  vqa_frame_metrics *__restrict__ res)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= pairs) return;
    double s = 0;
    for (int i = 0; i < FB_MAG_BLOCKS; i++) s += partials[(int64_t)p * FB_MAG_BLOCKS + i];
    res[p].flow_mag_mean = (p == 0 && !first_valid) ? 0.0 : s * inv_count;
}

// ---- launchers -------------------------------------------------------------------------------------
void launch_fb_blur(hipStream_t st, const uint8_t *gray, int pitch, int64_t plane_stride, int planes, int h, int w,
                    const fb_taps &T, const int32_t *cols, int nc, const int32_t *rows, int nr, float *tmp, float *out)
{
    if (!cols) nc = w;
    if (!rows) nr = h;
    hipLaunchKernelGGL(k_fb_blur_h, dim3((nc + 255) / 256, h, planes), dim3(256), 0, st, gray, pitch, plane_stride, h, w, T,
                       cols, nc, tmp);
    hipLaunchKernelGGL(k_fb_blur_v, dim3((nc + 255) / 256, nr, planes), dim3(256), 0, st, tmp, h, w, T, cols, nc, rows, out);
}

void launch_fb_resize(hipStream_t st, const float *src, int sh, int sw, int cn, float *dst, int dh, int dw, int images,
                      const fb_resize_tabs &T, float mul, bool apply_mul)
{
    dim3 grid((dw + 255) / 256, dh, images);
    if (cn == 1)
        hipLaunchKernelGGL(k_fb_resize<1>, grid, dim3(256), 0, st, src, sh, sw, dst, dh, dw, T.xofs, T.xa, T.yofs, T.yb,
                           T.mode, mul, (int)apply_mul);
    else
        hipLaunchKernelGGL(k_fb_resize<2>, grid, dim3(256), 0, st, src, sh, sw, dst, dh, dw, T.xofs, T.xa, T.yofs, T.yb,
                           T.mode, mul, (int)apply_mul);
}

void launch_fb_polyexp(hipStream_t st, const float *in, int planes, int h, int w, const fb_poly &C, float *out)
{
    dim3 grid((w + PE_TX - 1) / PE_TX, (h + PE_TY - 1) / PE_TY, planes);
    hipLaunchKernelGGL(k_fb_polyexp, grid, dim3(256), 0, st, in, h, w, C, out);
}

void launch_fb_update(hipStream_t st, const float *R, const float *flow, int pairs, int h, int w, float *M)
{
    dim3 grid((w + 255) / 256, h, pairs);
    hipLaunchKernelGGL(k_fb_update<0>, grid, dim3(256), 0, st, R, flow, h, w, M, 0, 0, nullptr, nullptr, nullptr, nullptr, 0.f);
}

// first rebuild of a level: coarse == nullptr -> zero flow, else the coarser level's flow upsampled in flight
void launch_fb_update_first(hipStream_t st, const float *R, const float *coarse, int ch, int cw, const fb_resize_tabs &T,
                            float mul, int pairs, int h, int w, float *M)
{
    dim3 grid((w + 255) / 256, h, pairs);
    if (!coarse)
        hipLaunchKernelGGL(k_fb_update<2>, grid, dim3(256), 0, st, R, nullptr, h, w, M, 0, 0, nullptr, nullptr, nullptr,
                           nullptr, 0.f);
    else
        hipLaunchKernelGGL(k_fb_update<1>, grid, dim3(256), 0, st, R, coarse, h, w, M, ch, cw, T.xofs, T.xa, T.yofs, T.yb, mul);
}

void launch_fb_blur_solve(hipStream_t st, const float *M, int pairs, int h, int w, float *flow)
{
    dim3 grid((w + BS_TX - 1) / BS_TX, (h + BS_TY - 1) / BS_TY, pairs);
    hipLaunchKernelGGL(k_fb_blur_solve, grid, dim3(BS_NT), 0, st, M, h, w, flow);
}

int fb_mag_blocks() { return FB_MAG_BLOCKS; }

void launch_fb_mag(hipStream_t st, const float *flow, int pairs, int h, int w, double *partials, bool first_valid,
                   vqa_frame_metrics *res)
{
    hipLaunchKernelGGL(k_fb_mag, dim3(FB_MAG_BLOCKS, pairs), dim3(256), 0, st, flow, (int64_t)h * w, partials);
    hipLaunchKernelGGL(k_fb_mag_finalize, dim3((pairs + 63) / 64), dim3(64), 0, st, partials, pairs,
                       1.0 / ((double)h * (double)w), (int)first_valid, res);
}

} // namespace vqa
