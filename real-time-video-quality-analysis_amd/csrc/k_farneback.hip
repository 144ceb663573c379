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
// in the flow there.  (The vertical box sums follow the oracle's order exactly; the horizontal ones differ from it at the
// 1e-16 level only.)
//
// Structure per pyramid level (coarse to fine), for a chunk of frame pairs whose planes stay resident (round 4):
//   k_fb_level3 / k_fb_level: float(gray) -> separable Gaussian (BORDER_REFLECT_101) -> resize to the level    (per plane,
//                   (2x2 mean when exactly halving), ONE kernel, only the samples the level reads                  shared by
//   k_fb_polyexp_march: 11x11 polynomial expansion -> 5 coefficient PLANES per gray plane                          both pairs)
//   k_fb_upflow   : the coarser level's flow upsampled and doubled (once per level; source patch through LDS)
//   k_fb_iter     : one flow iteration = bilinear warp of the second expansion by the flow -> 5 products per pixel
//                   (registers only) -> 15x15 box sums in double, OpenCV's running column sums -> regularised 2x2 solve
//                   (the last iteration of the finest level also sums |flow| for the mean magnitude)
// All of it is stencil work on fp32 planes: HBM / LDS / VALU, no MFMA.
// The kernels of rounds 2-3 (k_fb_blur_h / k_fb_blur_v / k_fb_resize<1> as three passes: the fallback for a level whose
// patch exceeds the LDS budget; k_fb_update + k_fb_blur_solve with the products through HBM: lab build only) are kept
// below next to what replaced them; LAB_NOTES.md L6 has the measurements.
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
// A workgroup takes FB_RB rows of its 256 columns: with one row each, a 1080p batch was ~280 000 workgroups of one
// output per thread and every one of these kernels ran at the dispatcher's rate (~220 us whatever the tap count).
constexpr int FB_RB = 16, FB_IL = 4;

// grid = (ceil(nc/256), ceil(h/FB_RB), planes)
__global__ __launch_bounds__(256) void k_fb_blur_h(const uint8_t *__restrict__ gray, int pitch, int64_t plane_stride,
                                                   int h, int w, fb_taps T, const int32_t *__restrict__ cols, int nc,
                                                   float *__restrict__ out)
{
    // the taps are indexed in a loop of run-time length: read from the by-value argument that is a private-memory
    // (scratch) access per tap; one LDS copy per workgroup instead
    __shared__ float kk[32];
    if (threadIdx.x < 32) kk[threadIdx.x] = T.k[threadIdx.x];
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nc) return;
    const int x = cols ? cols[i] : i;
    const int r = T.ksize >> 1;
    const bool interior = x >= r && x + r < w; // no reflection
    const int y1 = min((int)(blockIdx.y + 1) * FB_RB, h);
    const uint8_t *plane = gray + (int64_t)blockIdx.z * plane_stride;
    // FB_IL rows at a time: a thread with one row's three loads in flight left the kernel latency-bound at full occupancy
    for (int y = blockIdx.y * FB_RB; y < y1; y += FB_IL) {
        const uint8_t *row[FB_IL];
        float a[FB_IL];
#pragma unroll
        for (int j = 0; j < FB_IL; j++) {
            row[j] = plane + (int64_t)min(y + j, y1 - 1) * pitch;
            a[j] = (float)row[j][x] * kk[r];
        }
        if (interior) {
            for (int k = 1; k <= r; k++) {
                const float kf = kk[r + k];
#pragma unroll
                for (int j = 0; j < FB_IL; j++) a[j] += ((float)row[j][x - k] + (float)row[j][x + k]) * kf;
            }
        } else {
            for (int k = 1; k <= r; k++) {
                const int xl = fb_reflect101(x - k, w), xr = fb_reflect101(x + k, w);
                const float kf = kk[r + k];
#pragma unroll
                for (int j = 0; j < FB_IL; j++) a[j] += ((float)row[j][xl] + (float)row[j][xr]) * kf;
            }
        }
#pragma unroll
        for (int j = 0; j < FB_IL; j++)
            if (y + j < y1) out[((int64_t)blockIdx.z * h + y + j) * w + x] = a[j];
    }
}

// grid = (ceil(nc/256), ceil(nr/FB_RB), planes)
__global__ __launch_bounds__(256) void k_fb_blur_v(const float *__restrict__ in, int h, int w, fb_taps T,
                                                   const int32_t *__restrict__ cols, int nc,
                                                   const int32_t *__restrict__ rows, int nr, float *__restrict__ out)
{
    __shared__ float kk[32];
    if (threadIdx.x < 32) kk[threadIdx.x] = T.k[threadIdx.x];
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nc) return;
    const int x = cols ? cols[i] : i;
    const float *img = in + (int64_t)blockIdx.z * h * w;
    const int r = T.ksize >> 1;
    const int j1 = min((int)(blockIdx.y + 1) * FB_RB, nr);
    for (int j0 = blockIdx.y * FB_RB; j0 < j1; j0 += FB_IL) {
        int y[FB_IL];
        float a[FB_IL];
        bool inner = true;
#pragma unroll
        for (int j = 0; j < FB_IL; j++) {
            const int jj = min(j0 + j, j1 - 1);
            y[j] = rows ? rows[jj] : jj;
            inner = inner && y[j] >= r && y[j] + r < h;
            a[j] = img[(int64_t)y[j] * w + x] * kk[r];
        }
        if (inner) {
            for (int k = 1; k <= r; k++) {
                const float kf = kk[r + k];
#pragma unroll
                for (int j = 0; j < FB_IL; j++) {
                    const float *c = img + (int64_t)y[j] * w + x;
                    a[j] += (c[-(int64_t)k * w] + c[(int64_t)k * w]) * kf;
                }
            }
        } else {
            for (int k = 1; k <= r; k++) {
                const float kf = kk[r + k];
#pragma unroll
                for (int j = 0; j < FB_IL; j++)
                    a[j] += (img[(int64_t)fb_reflect101(y[j] - k, h) * w + x] + img[(int64_t)fb_reflect101(y[j] + k, h) * w + x]) * kf;
            }
        }
#pragma unroll
        for (int j = 0; j < FB_IL; j++)
            if (j0 + j < j1) out[((int64_t)blockIdx.z * h + y[j]) * w + x] = a[j];
    }
}

// ---- cv2.resize INTER_LINEAR on float data with CN interleaved channels; result times `mul` --------
// mode 0: bilinear with host-built tables; mode 1: exact 2x decimation (INTER_AREA fast path)
// grid = (ceil(dw/256), ceil(dh/FB_RB), images)
template <int CN>
__global__ __launch_bounds__(256) void k_fb_resize(const float *__restrict__ src, int sh, int sw, float *__restrict__ dst,
                                                   int dh, int dw, const int32_t *__restrict__ xofs,
                                                   const float *__restrict__ xa, const int32_t *__restrict__ yofs,
                                                   const float *__restrict__ yb, int mode, float mul, int apply_mul)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= dw) return;
    const float *s = src + (int64_t)blockIdx.z * sh * sw * CN;
    const int yend = min((int)(blockIdx.y + 1) * FB_RB, dh);
    if (mode == 1) {
#pragma unroll 4
        for (int y = blockIdx.y * FB_RB; y < yend; y++) {
            float *d = dst + (((int64_t)blockIdx.z * dh + y) * dw + x) * CN;
#pragma unroll
            for (int c = 0; c < CN; c++) {
                const float *p = s + ((int64_t)(2 * y) * sw + 2 * x) * CN + c;
                float v = (p[0] + p[CN] + p[(int64_t)sw * CN] + p[(int64_t)sw * CN + CN]) * 0.25f;
                if (apply_mul) v *= mul;
                d[c] = v;
            }
        }
        return;
    }
    const int x0 = xofs[x], x1 = min(x0 + 1, sw - 1);
    const float a0 = xa[2 * x], a1 = xa[2 * x + 1];
    if (CN == 2) {
        // flow fields: pixels are float2 (8-byte aligned: one load per corner, one store per pixel)
        const float2 *s2 = reinterpret_cast<const float2 *>(s);
        float2 *d2 = reinterpret_cast<float2 *>(dst) + (int64_t)blockIdx.z * dh * dw + x;
#pragma unroll 4
        for (int y = blockIdx.y * FB_RB; y < yend; y++) {
            const int y0 = min(max(yofs[y], 0), sh - 1), y1 = min(max(yofs[y] + 1, 0), sh - 1);
            const float b0 = yb[2 * y], b1 = yb[2 * y + 1];
            const float2 p00 = s2[(int64_t)y0 * sw + x0], p01 = s2[(int64_t)y0 * sw + x1];
            const float2 p10 = s2[(int64_t)y1 * sw + x0], p11 = s2[(int64_t)y1 * sw + x1];
            float2 v;
            {
                const float r0 = p00.x * a0 + p01.x * a1, r1 = p10.x * a0 + p11.x * a1;
                v.x = r0 * b0 + r1 * b1;
            }
            {
                const float r0 = p00.y * a0 + p01.y * a1, r1 = p10.y * a0 + p11.y * a1;
                v.y = r0 * b0 + r1 * b1;
            }
            if (apply_mul) { v.x *= mul; v.y *= mul; }
            d2[(int64_t)y * dw] = v;
        }
        return;
    }
#pragma unroll 4
    for (int y = blockIdx.y * FB_RB; y < yend; y++) {
        float *d = dst + (((int64_t)blockIdx.z * dh + y) * dw + x) * CN;
        const int y0 = min(max(yofs[y], 0), sh - 1), y1 = min(max(yofs[y] + 1, 0), sh - 1);
        const float b0 = yb[2 * y], b1 = yb[2 * y + 1];
#pragma unroll
        for (int c = 0; c < CN; c++) {
            const float r0 = s[((int64_t)y0 * sw + x0) * CN + c] * a0 + s[((int64_t)y0 * sw + x1) * CN + c] * a1;
            const float r1 = s[((int64_t)y1 * sw + x0) * CN + c] * a0 + s[((int64_t)y1 * sw + x1) * CN + c] * a1;
            float v = r0 * b0 + r1 * b1;
            if (apply_mul) v *= mul;
            d[c] = v;
        }
    }
}

// ---- flow upsample (cv2.resize INTER_LINEAR of a 2-channel field to a LARGER size, times mul) ---------------------
// k_fb_resize<2> gathers four float2 corners per output pixel from global memory: at 2x that is 16 loads for every 2 x 2
// output quad that shares 3 x 3 source pixels, and the texture path was its limit (0.28 ms per 32 fields of 1080p).  Here a
// workgroup stages the source patch of its 256 x 8 output tile in LDS with coalesced loads (upscale: at most 257 x 9
// source pixels) and the corners come from there.  Same tables, same expressions, same order as k_fb_resize<2>.
// grid = (ceil(dw / 256), ceil(dh / 8), fields), block = 256; requires dw >= sw and dh >= sh
constexpr int UF_ROWS = 8;

__global__ __launch_bounds__(256) void k_fb_upflow(const float *__restrict__ src, int sh, int sw, float *__restrict__ dst, int dh,
                                                   int dw, const int32_t *__restrict__ xofs, const float *__restrict__ xa,
                                                   const int32_t *__restrict__ yofs, const float *__restrict__ yb, float mul)
{
    __shared__ float2 tile[UF_ROWS + 1][257];
    const int t = threadIdx.x, X = blockIdx.x * 256 + t;
    const int Y0 = blockIdx.y * UF_ROWS, Y1 = min(Y0 + UF_ROWS, dh);
    const float2 *s2 = reinterpret_cast<const float2 *>(src) + (int64_t)blockIdx.z * sh * sw;
    // the tables are non-decreasing: the tile's patch is [first column's x0 .. last column's x1] x [first row's y0 .. last row's y1]
    const int Xl = min(blockIdx.x * 256 + 255, dw - 1);
    const int xlo = xofs[blockIdx.x * 256], xhi = min(xofs[Xl] + 1, sw - 1);
    const int ylo = min(max(yofs[Y0], 0), sh - 1), yhi = min(max(yofs[Y1 - 1] + 1, 0), sh - 1);
    const int ncols = xhi - xlo + 1, nrows = yhi - ylo + 1; // <= 257, <= UF_ROWS + 1 for an upscale
    for (int r = 0; r < nrows; r++)
        for (int c = t; c < ncols; c += 256) tile[r][c] = s2[(int64_t)(ylo + r) * sw + xlo + c];
    __syncthreads();
    if (X >= dw) return;
    const int x0 = xofs[X] - xlo, x1 = min(xofs[X] + 1, sw - 1) - xlo;
    const float a0 = xa[2 * X], a1 = xa[2 * X + 1];
    float2 *d2 = reinterpret_cast<float2 *>(dst) + (int64_t)blockIdx.z * dh * dw + X;
#pragma unroll 4
    for (int y = Y0; y < Y1; y++) {
        const int y0 = min(max(yofs[y], 0), sh - 1) - ylo, y1 = min(max(yofs[y] + 1, 0), sh - 1) - ylo;
        const float b0 = yb[2 * y], b1 = yb[2 * y + 1];
        const float2 p00 = tile[y0][x0], p01 = tile[y0][x1], p10 = tile[y1][x0], p11 = tile[y1][x1];
        float2 v;
        {
            const float r0 = p00.x * a0 + p01.x * a1, r1 = p10.x * a0 + p11.x * a1;
            v.x = r0 * b0 + r1 * b1;
        }
        {
            const float r0 = p00.y * a0 + p01.y * a1, r1 = p10.y * a0 + p11.y * a1;
            v.y = r0 * b0 + r1 * b1;
        }
        v.x *= mul;
        v.y *= mul;
        d2[(int64_t)y * dw] = v;
    }
}

// ---- the level image in one kernel: Gaussian blur of the u8 plane + cv2.resize to the level ---------------------
// Round 4.  The three kernels above (blur_h -> tmp, blur_v -> blur, resize -> level image) ran ~0.3 ms per level on a
// 33-plane 1080p chunk whatever the level, four levels per chunk: thousands of tiny latency-bound launches' worth of
// work.  Here a workgroup owns a tile of the LEVEL image and forms exactly the blurred samples its pixels read:
//   S = 2: the four bilinear taps (or the 2x2 mean of an exact halving) of each pixel sit at the sample columns
//          sx[2X], sx[2X+1] and sample rows sy[2Y], sy[2Y+1] (host tables: clamps applied); a tile is 32 x NSY/2 pixels;
//   S = 1: the finest level, no resize: one sample per pixel, a tile is 64 x NSY pixels.
// Stages (all in LDS): A the u8 patch around the tile's samples, addressed by VIRTUAL row / column so that
// BORDER_REFLECT_101 is resolved once here; B the horizontal pass at the 64 sample columns of every patch row; C the
// vertical pass at the NSY sample rows; D the resize arithmetic.  Every float expression is the one of
// k_fb_blur_h / k_fb_blur_v / k_fb_resize<1> in the same order, so the level image is bit-identical to theirs.
// grid = (ceil(lw / (64/S)), ceil(lh / (NSY/S)), planes), block = 256, dynamic LDS = fb_level_lds(...)
struct fb_level_args {
    const uint8_t *gray; int pitch; int64_t plane_stride; int h, w;   // source planes
    float *out; int lh, lw;                                           // level images, lh x lw floats per plane
    const int32_t *sx, *sy;                                           // S = 2: sample coordinates, 2 per level column / row
    const float *xa, *yb; int mode;                                   // S = 2: bilinear weights; mode 1 = 2x2 mean
    int cap_x, cap_y;                                                 // patch capacity (columns incl. 2r, rows incl. 2r)
    int tsx, tsy;                                                     // samples a tile really uses (<= 64, <= NSY; multiples of S):
                                                                      // 62 when r = 1, so that the patch is exactly 64 wide / high
};

static inline size_t fb_level_lds(int cap_x, int cap_y, int nsy)
{
    return (size_t)(((cap_x + 3) & ~3) + 4) * cap_y + sizeof(float) * 64 * cap_y + sizeof(float) * 64 * nsy + sizeof(float) * 32 +
           sizeof(int) * (64 + nsy);
}

template <int S, int NSY>
__global__ __launch_bounds__(256) void k_fb_level(fb_level_args A, fb_taps T)
{
    extern __shared__ __align__(16) unsigned char fl_lds[];
    const int px = ((A.cap_x + 3) & ~3) + 4;               // patch row pitch: + 4 for the dword-aligned staging below
    float *hb = reinterpret_cast<float *>(fl_lds);        // [cap_y][64]
    float *bt = hb + 64 * A.cap_y;                         // [NSY][64]
    float *kk = bt + 64 * NSY;                             // [32]
    int *sc = reinterpret_cast<int *>(kk + 32);            // [64] sample columns
    int *sr = sc + 64;                                     // [NSY] sample rows
    unsigned char *u8t = reinterpret_cast<unsigned char *>(sr + NSY); // [cap_y][px]
    const int t = threadIdx.x;
    const int X0 = blockIdx.x * (A.tsx / S), Y0 = blockIdx.y * (A.tsy / S);
    const int r = T.ksize >> 1;
    if (t < 32) kk[t] = T.k[t];
    if (t < 64) {
        const int i = min(t, A.tsx - 1); // the unused sample slots repeat the last one (the patch extent is sc[63] - sc[0])
        if (S == 1) sc[t] = min(X0 + i, A.w - 1);
        else sc[t] = A.sx[2 * min(X0 + (i >> 1), A.lw - 1) + (i & 1)];
    }
    if (t >= 64 && t < 64 + NSY) {
        const int j = min(t - 64, A.tsy - 1);
        if (S == 1) sr[t - 64] = min(Y0 + j, A.h - 1);
        else sr[t - 64] = A.sy[2 * min(Y0 + (j >> 1), A.lh - 1) + (j & 1)];
    }
    __syncthreads();
    const int xlo = sc[0] - r, ylo = sr[0] - r;
    const int xspan = sc[63] - sc[0] + 1 + 2 * r, yspan = sr[NSY - 1] - sr[0] + 1 + 2 * r; // <= cap_x, cap_y (host-checked)
    const uint8_t *plane = A.gray + (int64_t)blockIdx.z * A.plane_stride;
    // A: the patch, virtual coordinates -> reflected source coordinates.
    int xoff = 0; // where virtual column xlo sits in a patch row
    if (xlo >= 0 && xlo + xspan <= A.w && (A.pitch & 3) == 0 && (A.plane_stride & 3) == 0 && ((uintptr_t)A.gray & 3) == 0) {
        // no reflection along x (every tile but the frame's first and last column of tiles): aligned DWORD loads, 4 pixels
        // each - a fifth of the load instructions of the byte path below.  Rows still go through the reflection.
        xoff = xlo & 3;
        const int a0 = xlo - xoff, ndw = (xoff + xspan + 3) >> 2, total = yspan * ndw;
        const float inv = 1.0f / (float)ndw;
        uint32_t *u32t = reinterpret_cast<uint32_t *>(u8t);
        for (int i0 = t; i0 < total; i0 += 256 * 8) { // eight loads per thread in flight
            uint32_t v[8];
            int dst[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int i = min(i0 + 256 * u, total - 1);
                const int q = (int)(((float)i + 0.5f) * inv), d = i - q * ndw; // i / ndw, exact for i < 2^16 and ndw <= 128
                v[u] = *reinterpret_cast<const uint32_t *>(plane + (int64_t)fb_reflect101(ylo + q, A.h) * A.pitch + a0 + 4 * d);
                dst[u] = q * (px >> 2) + d;
            }
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (i0 + 256 * u < total) u32t[dst[u]] = v[u];
        }
    } else {
        // byte path.  Eight rows per thread in flight: with one load per iteration the stage was a chain of 20-80 dependent
        // global round trips per workgroup.
        for (int vx = t & 63; vx < xspan; vx += 64) {
            const uint8_t *col = plane + fb_reflect101(xlo + vx, A.w);
            for (int vy0 = t >> 6; vy0 < yspan; vy0 += 32) {
                unsigned char v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) v[u] = col[(int64_t)fb_reflect101(ylo + min(vy0 + 4 * u, yspan - 1), A.h) * A.pitch];
#pragma unroll
                for (int u = 0; u < 8; u++)
                    if (vy0 + 4 * u < yspan) u8t[(vy0 + 4 * u) * px + vx] = v[u];
            }
        }
    }
    __syncthreads();
    // B: horizontal pass at the sample columns, every patch row (four rows per thread in flight)
    {
        const int i = t & 63;
        const unsigned char *c0 = u8t + (sc[i] - xlo) + xoff;
        for (int vy0 = t >> 6; vy0 < yspan; vy0 += 16) {
            const unsigned char *c[4];
            float a[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                c[u] = c0 + min(vy0 + 4 * u, yspan - 1) * px;
                a[u] = (float)c[u][0] * kk[r];
            }
            for (int k = 1; k <= r; k++) {
                const float kf = kk[r + k];
#pragma unroll
                for (int u = 0; u < 4; u++) a[u] += ((float)c[u][-k] + (float)c[u][k]) * kf;
            }
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (vy0 + 4 * u < yspan) hb[(vy0 + 4 * u) * 64 + i] = a[u];
        }
    }
    __syncthreads();
    // C: vertical pass at the sample rows (four per thread in flight; NSY is a multiple of 16)
    {
        const int i = t & 63;
        for (int j0 = t >> 6; j0 < NSY; j0 += 16) {
            const float *c[4];
            float a[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                c[u] = hb + (sr[j0 + 4 * u] - ylo) * 64 + i;
                a[u] = c[u][0] * kk[r];
            }
            for (int k = 1; k <= r; k++) {
                const float kf = kk[r + k];
#pragma unroll
                for (int u = 0; u < 4; u++) a[u] += (c[u][-64 * k] + c[u][64 * k]) * kf;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int j = j0 + 4 * u;
                if (S == 1) {
                    const int X = X0 + i, Y = Y0 + j;
                    if (i < A.tsx && j < A.tsy && X < A.lw && Y < A.lh) A.out[((int64_t)blockIdx.z * A.lh + Y) * A.lw + X] = a[u];
                } else {
                    bt[j * 64 + i] = a[u];
                }
            }
        }
    }
    if (S == 1) return;
    __syncthreads();
    // D: cv2.resize on the four samples of each pixel
    {
        const int ix = t & 31, X = X0 + ix;
        for (int jy = t >> 5; jy < NSY / 2; jy += 8) {
            const int Y = Y0 + jy;
            if (2 * ix >= A.tsx || 2 * jy >= A.tsy || X >= A.lw || Y >= A.lh) continue;
            const float *p = bt + (2 * jy) * 64 + 2 * ix;
            float v;
            if (A.mode == 1) {
                v = (p[0] + p[1] + p[64] + p[65]) * 0.25f;
            } else {
                const float a0 = A.xa[2 * X], a1 = A.xa[2 * X + 1], b0 = A.yb[2 * Y], b1 = A.yb[2 * Y + 1];
                const float r0 = p[0] * a0 + p[1] * a1;
                const float r1 = p[64] * a0 + p[65] * a1;
                v = r0 * b0 + r1 * b1;
            }
            A.out[((int64_t)blockIdx.z * A.lh + Y) * A.lw + X] = v;
        }
    }
}

#ifndef FB_POLY_MARCH
#define FB_POLY_MARCH 1 // 1 = shipped: k_fb_polyexp_march; 0 = measurement build: the tile kernel of rounds 1-3
#endif
constexpr int PE_N = 5;

#if !FB_POLY_MARCH
// ---- polynomial expansion (FarnebackPolyExp, n = 5) ------------------------------------------------
// Tile of PE_TY rows x PE_TX columns per workgroup: the vertical pass (float) fills LDS for the tile's
// columns plus a 5-column replicated halo, the horizontal pass (double accumulators) reads it back.
constexpr int PE_TX = 64, PE_TY = 16;   // 16 rows per tile: the vertical pass re-reads 26 rows per 16 (4-row tiles: 14 per 4)
constexpr int PE_TILES = 2; // consecutive row tiles per workgroup

// grid = (ceil(w/PE_TX), ceil(h/(PE_TY * PE_TILES)), planes), block = 256
__global__ __launch_bounds__(256) void k_fb_polyexp(const float *__restrict__ in, int h, int w, fb_poly C,
                                                    float *__restrict__ out)
{
    __shared__ float row[PE_TY][PE_TX + 2 * PE_N][3];
    const float *img = in + (int64_t)blockIdx.z * h * w;
    const int x0 = blockIdx.x * PE_TX;
    const float *g = C.g + PE_N, *xg = C.xg + PE_N, *xxg = C.xxg + PE_N;
    const int64_t P = (int64_t)h * w;
    for (int tile = 0; tile < PE_TILES; tile++) {
        const int y0 = (blockIdx.y * PE_TILES + tile) * PE_TY;
        if (y0 >= h) break;
        if (tile) __syncthreads(); // the previous tile's readers are done
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
        const int tx = threadIdx.x % PE_TX, x = x0 + tx;
        if (x >= w) continue;
#pragma unroll 2
        for (int ty = threadIdx.x / PE_TX; ty < PE_TY; ty += 256 / PE_TX) {
            const int y = y0 + ty;
            if (y >= h) break;
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
            // the five coefficients go to five PLANES (channel-major): the flow iteration reads them one column per lane,
            // and planar the loads of a row are contiguous and the bilinear corners x1, x1 + 1 are one 8-byte load
            float *d = out + (int64_t)blockIdx.z * P * 5 + (int64_t)y * w + x;
            d[P] = (float)(b2 * C.ig11);
            d[0] = (float)(b3 * C.ig11);
            d[3 * P] = (float)(b1 * C.ig03 + b4 * C.ig33);
            d[2 * P] = (float)(b1 * C.ig03 + b5 * C.ig33);
            d[4 * P] = (float)(b6 * C.ig55);
        }
    }
}

#endif // !FB_POLY_MARCH

#if FB_POLY_MARCH
// ---- polynomial expansion, marching form (round 4) ---------------------------------------------------------------
// The tile kernel above re-reads 26 rows per 16 (and clamps 11 addresses per LDS entry); here a workgroup owns PM_NT
// columns (PM_OUT = PM_NT - 10 outputs + the 5-column replicated halo either side) of a row strip and marches down it:
// thread = column, the 11 rows of its vertical taps sit in a register ring (ONE load per row and thread), the three
// vertical sums of a row go through LDS (double-buffered: one barrier per row), the horizontal pass and the 6x6 solve
// are the tile kernel's expressions in the same order, so the coefficient planes are bit-identical to its.
// grid = (ncb * ns, planes), block = PM_NT; QS = rows per strip
constexpr int PM_NT = 256, PM_OUT = PM_NT - 2 * PE_N;

__global__ __launch_bounds__(PM_NT) void k_fb_polyexp_march(const float *__restrict__ in, int h, int w, fb_poly C,
                                                            float *__restrict__ out, int ncb, int QS)
{
    __shared__ float rowb[2][3][PM_NT]; // (one float4 per column - a 16-byte LDS access per tap instead of three - measured no faster)
    const int t = threadIdx.x;
    const int cb = blockIdx.x % ncb, sb = blockIdx.x / ncb;
    const int x0 = cb * PM_OUT, ys = sb * QS, yend = min(ys + QS, h);
    const int xx = min(max(x0 - PE_N + t, 0), w - 1); // the horizontal pass replicates the border triples
    const float *col = in + (int64_t)blockIdx.z * h * w + xx;
    const float *g = C.g + PE_N, *xg = C.xg + PE_N, *xxg = C.xxg + PE_N;
    const int64_t P = (int64_t)h * w;
    const int xo = x0 + t - PE_N;                      // this thread's output column (when 5 <= t < PM_NT - 5)
    const bool writer = t >= PE_N && t < PM_NT - PE_N && xo < w;
    float *dcol = out + (int64_t)blockIdx.z * P * 5 + xo;
    auto ld = [&](int y) { return col[(int64_t)min(max(y, 0), h - 1) * w]; };
    // ring slot of row y: (y - (ys - 5)) mod 11; the march below is unrolled by 11 so that the slots are static
    float win[11];
#pragma unroll
    for (int k = 0; k < 10; k++) win[k] = ld(ys - PE_N + k);  // rows ys-5 .. ys+4
    float nxt = ld(ys + PE_N);                                 // row ys+5, the entering row of output row ys
    int n = 0;
    for (int base = 0; ys + base < yend; base += 11) {
#pragma unroll
        for (int i = 0; i < 11; i++) {
            const int y = ys + base + i;
            if (y < yend) { // wave-uniform
                win[(i + 10) % 11] = nxt;       // row y + 5
                nxt = ld(y + PE_N + 1);         // one row ahead of its use
                // vertical pass: the tile kernel's expressions (a = row y - k, b = row y + k, both clamped by ld)
                const float c0 = win[(i + 5) % 11];
                float t0 = c0 * g[0], t1 = 0.f, t2 = 0.f;
#pragma unroll
                for (int k = 1; k <= PE_N; k++) {
                    const float a = win[(i + 5 - k) % 11], b = win[(i + 5 + k) % 11];
                    const float p = a + b;
                    t0 = t0 + g[k] * p;
                    t1 = t1 + xg[k] * (b - a);
                    t2 = t2 + xxg[k] * p;
                }
                float(*rb)[PM_NT] = rowb[n & 1];
                rb[0][t] = t0; rb[1][t] = t1; rb[2][t] = t2;
                __syncthreads();
                if (writer) {
                    double b1 = rb[0][t] * g[0], b2 = 0, b3 = rb[1][t] * g[0], b4 = 0, b5 = rb[2][t] * g[0], b6 = 0;
#pragma unroll
                    for (int k = 1; k <= PE_N; k++) {
                        const double tg = rb[0][t + k] + rb[0][t - k];
                        b1 += tg * g[k];
                        b4 += tg * xxg[k];
                        b2 += (rb[0][t + k] - rb[0][t - k]) * xg[k];
                        b3 += (rb[1][t + k] + rb[1][t - k]) * g[k];
                        b6 += (rb[1][t + k] - rb[1][t - k]) * xg[k];
                        b5 += (rb[2][t + k] + rb[2][t - k]) * g[k];
                    }
                    float *d = dcol + (int64_t)y * w;
                    d[P] = (float)(b2 * C.ig11);
                    d[0] = (float)(b3 * C.ig11);
                    d[3 * P] = (float)(b1 * C.ig03 + b4 * C.ig33);
                    d[2 * P] = (float)(b1 * C.ig03 + b5 * C.ig33);
                    d[4 * P] = (float)(b6 * C.ig55);
                }
                n++;
            }
        }
    }
}

#endif // FB_POLY_MARCH

// ---- FarnebackUpdateMatrices: pair p uses expansions of planes p and p + 1 -------------------------
// SRC: where the flow comes from.  0 = the level's flow field; 1 = the coarser level's flow, resized
// (INTER_LINEAR, the k_fb_resize<2> arithmetic) and doubled on the fly — the first rebuild of a level is its
// only reader, so the upsampled field is never written; 2 = zero (coarsest level).
template <int SRC>
__device__ __forceinline__ void fb_flow_at(const float *__restrict__ flow /* the pair's field */, int w, int x, int y, int ch,
                                           int cw, const int32_t *__restrict__ xofs, const float *__restrict__ xa,
                                           const int32_t *__restrict__ yofs, const float *__restrict__ yb, float mul,
                                           float &dx, float &dy)
{
    dx = 0.f;
    dy = 0.f;
    if (SRC == 0) {
        const float2 f = *reinterpret_cast<const float2 *>(reinterpret_cast<const char *>(flow) + (uint32_t)(y * w + x) * 8u);
        dx = f.x;
        dy = f.y;
    } else if (SRC == 1) {
        const int x0 = xofs[x], x1c = min(x0 + 1, cw - 1);
        const int y0 = min(max(yofs[y], 0), ch - 1), y1c = min(max(yofs[y] + 1, 0), ch - 1);
        const float a0 = xa[2 * x], a1 = xa[2 * x + 1], b0 = yb[2 * y], b1 = yb[2 * y + 1];
        float v[2];
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const float q0 = flow[((int64_t)y0 * cw + x0) * 2 + c] * a0 + flow[((int64_t)y0 * cw + x1c) * 2 + c] * a1;
            const float q1 = flow[((int64_t)y1c * cw + x0) * 2 + c] * a0 + flow[((int64_t)y1c * cw + x1c) * 2 + c] * a1;
            v[c] = (q0 * b0 + q1 * b1) * mul;
        }
        dx = v[0];
        dy = v[1];
    }
}

// the five products of pixel (x, y) under the displacement (dx, dy): R0 / R1 = the expansions of the pair's two planes,
// five coefficient planes of P = h * w floats each
struct __attribute__((packed, aligned(4))) fb_f2 { float a, b; }; // two adjacent floats, 4-byte aligned: one global_load_dwordx2

// Loads at (uniform base) + (32-bit byte offset): the base stays in scalar registers and the lane's offset is ONE
// register (global_load ... v_off, s[base]); with 64-bit per-lane pointers the march spent 18 vector instructions per
// row on address arithmetic.  Offsets fit: h * w <= 2^28 pixels (include/vqa.h), 8 bytes per pixel at most.
__device__ __forceinline__ float fb_ld(const float *base, uint32_t off) { return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + off); }
__device__ __forceinline__ fb_f2 fb_ld2(const float *base, uint32_t off) { return *reinterpret_cast<const fb_f2 *>(reinterpret_cast<const char *>(base) + off); }

__device__ __forceinline__ void fb_products(const float *__restrict__ R0, const float *__restrict__ R1, int h, int w, int x,
                                            int y, float dx, float dy, float m[5])
{
    const int64_t P = (int64_t)h * w;
    const uint32_t o0 = (uint32_t)(y * w + x) * 4u;
    const float r00 = fb_ld(R0, o0), r01 = fb_ld(R0 + P, o0), r02 = fb_ld(R0 + 2 * P, o0), r03 = fb_ld(R0 + 3 * P, o0),
                r04 = fb_ld(R0 + 4 * P, o0);
    float fx = x + dx, fy = y + dy;
    const int x1 = (int)floorf(fx), y1 = (int)floorf(fy);
    float r2, r3, r4, r5, r6;
    fx -= x1; fy -= y1;
    if ((unsigned)x1 < (unsigned)(w - 1) && (unsigned)y1 < (unsigned)(h - 1)) {
        const uint32_t o1 = (uint32_t)(y1 * w + x1) * 4u, o2 = o1 + (uint32_t)w * 4u;
        const float a00 = (1.f - fx) * (1.f - fy), a01 = fx * (1.f - fy), a10 = (1.f - fx) * fy, a11 = fx * fy;
        float v[5];
#pragma unroll
        for (int c = 0; c < 5; c++) {
            const fb_f2 t = fb_ld2(R1 + c * P, o1), b = fb_ld2(R1 + c * P, o2);
            v[c] = a00 * t.a + a01 * t.b + a10 * b.a + a11 * b.b;
        }
        r2 = v[0]; r3 = v[1];
        r4 = (r02 + v[2]) * 0.5f;
        r5 = (r03 + v[3]) * 0.5f;
        r6 = (r04 + v[4]) * 0.25f;
    } else {
        r2 = r3 = 0.f;
        r4 = r02; r5 = r03; r6 = r04 * 0.5f;
    }
    r2 = (r00 - r2) * 0.5f;
    r3 = (r01 - r3) * 0.5f;
    r2 += r4 * dy + r6 * dx;
    r3 += r6 * dy + r5 * dx;
    if ((unsigned)(x - 5) >= (unsigned)(w - 10) || (unsigned)(y - 5) >= (unsigned)(h - 10)) {
        const float border[5] = {0.14f, 0.14f, 0.4472f, 0.4472f, 0.4472f};
        const float scale = (x < 5 ? border[x] : 1.f) * (x >= w - 5 ? border[w - x - 1] : 1.f) *
                            (y < 5 ? border[y] : 1.f) * (y >= h - 5 ? border[h - y - 1] : 1.f);
        r2 *= scale; r3 *= scale; r4 *= scale; r5 *= scale; r6 *= scale;
    }
    m[0] = r4 * r4 + r6 * r6;
    m[1] = (r4 + r5) * r6;
    m[2] = r5 * r5 + r6 * r6;
    m[3] = r4 * r2 + r6 * r3;
    m[4] = r6 * r2 + r5 * r3;
}

constexpr int BS_M = 7; // the 15x15 box of FarnebackUpdateFlow_Blur (winsize 15)

#ifdef VQA_AB_VARIANTS // rounds 2-3: products written to HBM by one kernel, box-summed and solved by a second (lab build)
// grid = (ceil(w/256), h, pairs)
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
    const float *fp = SRC == 0 ? flow + (int64_t)blockIdx.z * P * 2 : SRC == 1 ? flow + (int64_t)blockIdx.z * ch * cw * 2 : nullptr;
    float dx, dy, m[5];
    fb_flow_at<SRC>(fp, w, x, y, ch, cw, xofs, xa, yofs, yb, mul, dx, dy);
    fb_products(R0, R1, h, w, x, y, dx, dy, m);
    float *o = M + ((int64_t)blockIdx.z * P + (int64_t)y * w + x) * 5;
#pragma unroll
    for (int c = 0; c < 5; c++) o[c] = m[c];
}

// ---- FarnebackUpdateFlow_Blur: 15x15 box sums of the products + 2x2 solve --------------------------
// Tile BS_TY x BS_TX outputs per workgroup.  The tile's products (+-7 halo, rows and columns clamped =
// replicated border) are staged in LDS once; stage 1 slides a 15-wide window along every (row, channel)
// in double, stage 2 slides a 15-high window down every (column, channel), stage 3 scales and solves.
// Sums are carried in double as in OpenCV; the sliding order differs from the oracle's tap order only at
// the 1e-16 level, far below what the regularised solve can amplify.
constexpr int BS_TX = 32, BS_TY = 16;
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

#endif // VQA_AB_VARIANTS

// ---- one flow iteration, fused: FarnebackUpdateMatrices + FarnebackUpdateFlow_Blur -----------------------------
// Round 4.  The two-kernel form above wrote the five products (20 B/pixel) to HBM and read them back with a 2.7x
// halo factor, and its sliding sums ran on 15 % of a workgroup's threads.  Here a workgroup owns FI_NT columns
// (FI_OUT = FI_NT - 14 output columns + the 7-column halo either side) of a row strip and MARCHES down it:
//   * thread = one column: it forms the products of the entering row (y + 7) in registers, keeps the last 16 rows of
//     them in a register ring, and carries OpenCV's running column sum  vsum += float(M[y+7] - M[y-8])  in double -
//     the same float difference, in the same order, as optflowgf.cpp / oracle fb_blur_solve - between RESTARTS: at every
//     output row that is a multiple of FI_RESTART (16: once per turn of the ring) below the top the sum is formed afresh as the plain sum of the 15
//     rows in the window (they are in the ring), in ascending row order.  That is exactly what a strip that STARTS at
//     that row gets from its priming, and strips start only at multiples of FI_RESTART - so the flow field is the same
//     bits whatever the number of strips, i.e. whatever the number of pairs in the launch (round 6; rounds 4-5 restarted
//     only where a strip began, so the field followed the batch size in its last bits).  Against OpenCV's whole-frame
//     running sum the restart removes float-difference drift: ~1e-7 on the column sums, far inside the 1e-4 bar;
//   * FI_R rows of column sums go through LDS per step; a group of FI_R lanes takes FI_R adjacent output columns of one
//     row each: FI_R + 14 doubles -> the 15-column window sum of the first and, sliding by one column at a time, of
//     the others ((FI_R + 14) / FI_R LDS reads and (14 + 2 (FI_R - 1)) / FI_R additions per output and channel: 4.5 and
//     5 for FI_R = 4, 8 and 8.5 for 2); then the regularised 2x2 solve and the flow store.
// The horizontal window is summed left to right per group, not slid along the whole row as OpenCV does: that
// differs at the 1e-16 level only.  Products are never written: per pixel the iteration reads 20 B (R0) + 20 B (R1,
// gathered) + 8 B (flow) and writes 8 B.
// grid = (ncb * ns, pairs), block = FI_NT; QS = rows per strip, a multiple of FI_RESTART (and so of the 16 static ring slots)
#ifndef FI_THREADS
#define FI_THREADS 256 // columns per workgroup (256 = shipped; 64 = measurement build: one free-running wave per workgroup)
#endif
constexpr int FI_NT = FI_THREADS, FI_OUT = FI_NT - 14;
#ifndef FB_PROBE
#define FB_PROBE 0 // measurement builds only (scripts/build_probes.sh FB_PROBE 2 ..): 2 = no products (constant instead); round 6's
                   // A/B of the batch-independent form: 3 = no restart code, 4 = magnitudes summed in double as in rounds 4-5,
                   // 5 = both = rounds 4-5's kernel (-DFI_RESTART_ROWS=64: the first, slower form of the restart)
#endif
#ifndef FI_RESTART_ROWS
#define FI_RESTART_ROWS 16 // 16 = shipped (= the 16 static ring slots: every loop iteration of the march restarts); 64: measurement build
#endif
constexpr int FI_RESTART = FI_RESTART_ROWS;
#ifndef FI_ROWS
#define FI_ROWS 2  // rows per LDS round trip (2 = shipped; 4 = measurement build: 222 VGPRs, two waves per SIMD, 1.5x slower)
#endif
constexpr int FI_R = FI_ROWS;
// LDS: vs[channel][row of the step][column] in doubles.  The lanes of a group read the SAME columns of different rows
// with ds_read_b128, so the rows of a step are staggered: two rows per step - 128 bytes apart modulo 256 (16 lanes = 8 pairs
// cover 128 bytes of row 0 and the other 128 of row 1); four rows - the 16-byte slots that the groups' own 32-byte spacing
// leaves free: row offsets 0, 16, 128, 144 bytes modulo 256.
constexpr int FI_ROWB = ((FI_NT + 5) * 8 + 255) / 256 * 256;  // bytes per row: a multiple of 256 >= (FI_NT + 3 + 2) * 8  (2304 for 256 columns)
constexpr int FI_CHB = FI_R * FI_ROWB + 256;                // bytes per channel
__device__ __forceinline__ constexpr int fi_row_off(int q)
{
    return FI_R == 2 ? q * (FI_ROWB + 128) : q * FI_ROWB + (q & 1) * 16 + (q >> 1) * 128;
}

// SRC = 0: the displacement comes from the level's flow field `fin`; SRC = 2: zero (coarsest level, fin unused).
// MAG: the last iteration of the finest level also sums |flow| of what it stores into mag[pair][workgroup] - the separate
// magnitude pass re-read the whole field for that.  The float magnitudes are added in 2^-28 fixed point: integer sums are
// associative, so the mean does not care how the field was cut into workgroups (round 6; doubles before).
template <int SRC, bool MAG>
__global__ __launch_bounds__(FI_NT) void k_fb_iter(const float *__restrict__ R, const float *__restrict__ fin, int h, int w,
                                                   float *__restrict__ fout, int ncb, int QS, double *__restrict__ mag)
{
    static_assert(SRC == 0 || SRC == 2, "the in-flight upsample of rounds 2-3 (SRC = 1) lives in the lab build's k_fb_update only");
    static_assert(FI_R == 2 || FI_R == 4, "rows per step = lanes per group = adjacent output columns per lane");
    // two buffers, alternating by step: ONE barrier per step (a step's stores go to the buffer read two steps ago, and every
    // thread passed the barrier in between only after finishing those reads; 16 / FI_R steps per loop iteration is even)
    __shared__ __align__(16) unsigned char vsb[2][5 * FI_CHB];
    __shared__ unsigned long long mag_red[FI_NT / 64];
    long long macc = 0;
#if FB_PROBE == 4 || FB_PROBE == 5
    double maccd = 0;
#endif
    const int t = threadIdx.x;
    const int cb = blockIdx.x % ncb, sb = blockIdx.x / ncb;
    const int x0 = cb * FI_OUT, ys = sb * QS;
    const int yend = min(ys + QS, h);
    const int64_t P = (int64_t)h * w;
    const float *R0 = R + (int64_t)blockIdx.y * P * 5, *R1 = R0 + P * 5;
    const float *fp = SRC == 0 ? fin + (int64_t)blockIdx.y * P * 2 : nullptr;
    float *fo = fout + (int64_t)blockIdx.y * P * 2;
    const int xx = min(max(x0 - BS_M + t, 0), w - 1); // replicated border: the clamped column's products ARE the border's
    float ring[16][5];
#pragma unroll
    for (int s = 0; s < 16; s++)
#pragma unroll
        for (int c = 0; c < 5; c++) ring[s][c] = 0.f;
    double vsum[5] = {0., 0., 0., 0., 0.};
    // horizontal role: output columns k0 .. k0 + FI_R - 1 (relative to x0) of row (t % FI_R) of the step
    const int k0 = t & ~(FI_R - 1), hrow = t & (FI_R - 1);
    const int vrow = (FI_R == 2 ? hrow * (FI_ROWB + 128) : hrow * FI_ROWB + (hrow & 1) * 16 + (hrow >> 1) * 128) + k0 * 8;
    bool outk[FI_R];
#pragma unroll
    for (int j = 0; j < FI_R; j++) outk[j] = k0 + j < FI_OUT && x0 + k0 + j < w;

    // Row gi of the strip's march: output row y = ys - 16 + gi, entering product row rho = y + 7 (clamped = replicated).
    // gi = 0 is a dummy (keeps the steps aligned with the 16-slot ring), gi = 1..15 prime the column sums with
    // the rows ys-8 .. ys+6 (the ring is zero there, so the difference form adds the plain values), gi >= 16 emit.
    // The displacement of a step's rows is fetched one step ahead: the gathers of R1 depend on it, and the two
    // dependent round trips per step were the kernel's critical path.
    auto row_of = [&](int y) { return min(max(y + BS_M, 0), h - 1); };
    float dxn[FI_R], dyn[FI_R];
#pragma unroll
    for (int q = 0; q < FI_R; q++) fb_flow_at<SRC>(fp, w, xx, row_of(ys - 16 + q), 0, 0, nullptr, nullptr, nullptr, nullptr, 0.f, dxn[q], dyn[q]);
    for (int base = 0;; base += 16) {
        const int yb0 = ys - 16 + base;
        if (yb0 >= yend) break;
#pragma unroll
        for (int i = 0; i < 16; i += FI_R) {
            float dxc[FI_R], dyc[FI_R], cur[FI_R][5];
#pragma unroll
            for (int q = 0; q < FI_R; q++) { dxc[q] = dxn[q]; dyc[q] = dyn[q]; }
#pragma unroll
            for (int q = 0; q < FI_R; q++) {
#if FB_PROBE == 2
                for (int c = 0; c < 5; c++) cur[q][c] = dxc[q] + dyc[q] * c;
#else
                fb_products(R0, R1, h, w, xx, row_of(yb0 + i + q), dxc[q], dyc[q], cur[q]);
#endif
            }
#pragma unroll
            for (int q = 0; q < FI_R; q++)
                fb_flow_at<SRC>(fp, w, xx, row_of(yb0 + i + FI_R + q), 0, 0, nullptr, nullptr, nullptr, nullptr, 0.f, dxn[q], dyn[q]);
            if (FB_PROBE != 3 && FB_PROBE != 5 && i == 0 && base > 16 && yb0 > 0 && (yb0 & (FI_RESTART - 1)) == 0) {
                // RESTART (workgroup-uniform): output row yb0 is a multiple of FI_RESTART inside this strip.  The ring holds the
                // product rows yb0 - 9 .. yb0 + 6; rows yb0 - 8 .. yb0 + 6 sit in slots 8 .. 15, 0 .. 6 (slot 7, row yb0 - 9, is
                // about to be overwritten).  Their plain sum, ascending, is what the priming of a strip starting here leaves.
#pragma unroll
                for (int c = 0; c < 5; c++) {
                    double a = 0.;
#pragma unroll
                    for (int m = 0; m < 15; m++) a += (double)ring[(8 + m) & 15][c];
                    vsum[c] = a;
                }
            }
#pragma unroll
            for (int q = 0; q < FI_R; q++) {
                const int y = yb0 + i + q, rho = y + BS_M;
                const bool dummy = i + q == 0 && base == 0;
#pragma unroll
                for (int c = 0; c < 5; c++) {
                    float add = cur[q][c] - ring[(i + q + 8) & 15][c]; // leaving row y - 8 (zero while priming)
                    // top of the frame, OpenCV's start: vsum = float(M[0] * 9) + M[1] + .. + M[6]; the rows -8 .. -1 are M[0]
                    if (y < 0) add = rho == -8 ? cur[q][c] * 9.f : (rho <= 0 ? 0.f : cur[q][c]);
                    if (dummy) add = 0.f;
                    vsum[c] += (double)add;
                    ring[(i + q + 7) & 15][c] = dummy ? 0.f : cur[q][c];
                    *reinterpret_cast<double *>(vsb[(i / FI_R) & 1] + c * FI_CHB + fi_row_off(q) + t * 8) = vsum[c];
                }
            }
            if (base > 0) {
                __syncthreads();
                const int yo = yb0 + i + hrow;
                if (yo < yend && outk[0]) {
                    double sw[FI_R][5];
#pragma unroll
                    for (int c = 0; c < 5; c++) {
                        const double2 *p = reinterpret_cast<const double2 *>(vsb[(i / FI_R) & 1] + vrow + c * FI_CHB);
                        constexpr int NV = (FI_R + 14 + 1) / 2;
                        double2 v[NV];
#pragma unroll
                        for (int k = 0; k < NV; k++) v[k] = p[k];
                        auto at = [&](int k) { return (k & 1) ? v[k >> 1].y : v[k >> 1].x; };
                        double a = at(0);
#pragma unroll
                        for (int k = 1; k < 15; k++) a += at(k);
                        sw[0][c] = a;
#pragma unroll
                        for (int j = 1; j < FI_R; j++) {
                            a = a + (at(14 + j) - at(j - 1));
                            sw[j][c] = a;
                        }
                    }
                    const double scale = 1. / 225.;
                    float *f = fo + ((int64_t)yo * w + x0 + k0) * 2;
#pragma unroll
                    for (int j = 0; j < FI_R; j++) {
                        if (!outk[j]) continue;
                        const double g11 = sw[j][0] * scale, g12 = sw[j][1] * scale, g22 = sw[j][2] * scale, h1 = sw[j][3] * scale,
                                     h2 = sw[j][4] * scale;
                        const double idet = 1. / (g11 * g22 - g12 * g12 + 1e-3);
                        const float fx = (float)((g11 * h2 - g12 * h1) * idet), fy = (float)((g22 * h1 - g12 * h2) * idet);
                        f[2 * j] = fx;
                        f[2 * j + 1] = fy;
#if FB_PROBE == 4 || FB_PROBE == 5
                        if (MAG) maccd += (double)sqrtf(fx * fx + fy * fy);
#else
                        if (MAG) macc += __float2ll_rn(sqrtf(fx * fx + fy * fy) * 268435456.f); // 2^28
#endif
                    }
                }
            }
        }
    }
#if FB_PROBE == 4 || FB_PROBE == 5
    macc = __double2ll_rn(maccd * 268435456.0); // (measurement build: the double chain's total, handed on in the shipped format)
#endif
    if (MAG) {
        const unsigned long long tot = block_sum_u64((unsigned long long)macc, mag_red); // (two's complement: signed sums wrap right)
        if (t == 0) reinterpret_cast<unsigned long long *>(mag)[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = tot;
    }
}

// ---- mean magnitude --------------------------------------------------------------------------------
// partials[pair][block] (left by the last k_fb_iter; in the lab build's two-kernel form by k_fb_mag), then a fixed-order
// finalize (bit-reproducible)
#ifdef VQA_AB_VARIANTS
// grid = (FB_MAG_BLOCKS, pairs)
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
#endif

// FIXED: the partials are 2^-28 fixed-point integer sums (the fused iteration), else doubles (the lab build's k_fb_mag)
template <bool FIXED>
__global__ void k_fb_mag_finalize(const double *__restrict__ partials, int nblk, int pairs, double inv_count, int first_valid,
                                  vqa_frame_metrics *__restrict__ res)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= pairs) return;
    double s = 0;
    if (FIXED) {
        long long t = 0;
        const long long *q = reinterpret_cast<const long long *>(partials);
        for (int i = 0; i < nblk; i++) t += q[(int64_t)p * nblk + i];
        s = (double)t * (1.0 / 268435456.0);
    } else {
        for (int i = 0; i < nblk; i++) s += partials[(int64_t)p * nblk + i];
    }
    res[p].flow_mag_mean = (p == 0 && !first_valid) ? 0.0 : s * inv_count;
}

// ---- launchers -------------------------------------------------------------------------------------
void launch_fb_blur(hipStream_t st, const uint8_t *gray, int pitch, int64_t plane_stride, int planes, int h, int w,
                    const fb_taps &T, const int32_t *cols, int nc, const int32_t *rows, int nr, float *tmp, float *out)
{
    if (!cols) nc = w;
    if (!rows) nr = h;
    hipLaunchKernelGGL(k_fb_blur_h, dim3((nc + 255) / 256, (h + FB_RB - 1) / FB_RB, planes), dim3(256), 0, st, gray, pitch,
                       plane_stride, h, w, T, cols, nc, tmp);
    hipLaunchKernelGGL(k_fb_blur_v, dim3((nc + 255) / 256, (nr + FB_RB - 1) / FB_RB, planes), dim3(256), 0, st, tmp, h, w, T, cols,
                       nc, rows, nr, out);
}

// ---- the level image of the two finest levels: 3-tap blur (+ exact halving), marching ----------------------------
// ksize 3 covers the finest level (no resize, S = 1) and the first halving (sigma 0.5, INTER_AREA's 2x2 mean, S = 2) of
// a 0.5 pyramid - 15/16 of the pyramid's pixels.  thread = one level column marching down a row strip: the horizontal
// pass of the entering source row(s) straight from the u8 plane (3 or 4 byte loads, L1), the previous ones in
// registers, no LDS, no barrier.  The expressions are k_fb_blur_h / k_fb_blur_v / k_fb_resize<1> (mode 1)'s, in order.
// grid = (ceil(lw / 256), ns, planes), block = 256; QS = level rows per strip
template <int S>
__global__ __launch_bounds__(256) void k_fb_level3(const uint8_t *__restrict__ gray, int pitch, int64_t plane_stride, int h, int w,
                                                   float k1, float k2, float *__restrict__ out, int lh, int lw, int QS)
{
    const int X = blockIdx.x * 256 + threadIdx.x;
    if (X >= lw) return;
    const uint8_t *plane = gray + (int64_t)blockIdx.z * plane_stride;
    float *o = out + (int64_t)blockIdx.z * lh * lw + X;
    const int Y0 = blockIdx.y * QS, Y1 = min(Y0 + QS, lh);
    if (S == 1) {
        const int xl = fb_reflect101(X - 1, w), xr = fb_reflect101(X + 1, w);
        auto hb = [&](int y) {
            const uint8_t *row = plane + (int64_t)fb_reflect101(y, h) * pitch;
            float a = (float)row[X] * k1;
            a += ((float)row[xl] + (float)row[xr]) * k2;
            return a;
        };
        float hm = hb(Y0 - 1), h0 = hb(Y0);
        for (int y = Y0; y < Y1; y += 4) { // four rows in flight
            float hn[4];
#pragma unroll
            for (int u = 0; u < 4; u++) hn[u] = hb(min(y + u, Y1 - 1) + 1);
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (y + u < Y1) {
                    float b = h0 * k1;
                    b += (hm + hn[u]) * k2;
                    o[(int64_t)(y + u) * lw] = b;
                    hm = h0;
                    h0 = hn[u];
                }
            }
        }
    } else {
        // source columns 2X, 2X + 1 (w = 2 lw exactly) and their reflected neighbours
        const int c0 = 2 * X, c1 = 2 * X + 1, cl = fb_reflect101(c0 - 1, w), cr = fb_reflect101(c1 + 1, w);
        auto hb2 = [&](int y, float &a0, float &a1) {
            const uint8_t *row = plane + (int64_t)fb_reflect101(y, h) * pitch;
            const float pl = (float)row[cl], p0 = (float)row[c0], p1 = (float)row[c1], pr = (float)row[cr];
            a0 = p0 * k1;
            a0 += (pl + p1) * k2;
            a1 = p1 * k1;
            a1 += (p0 + pr) * k2;
        };
        float am0, am1, a00, a01; // source rows 2Y - 1 and 2Y of the current level row
        hb2(2 * Y0 - 1, am0, am1);
        hb2(2 * Y0, a00, a01);
        for (int Y = Y0; Y < Y1; Y += 2) { // two level rows = four source rows in flight
            float n0[4], n1[4];
#pragma unroll
            for (int u = 0; u < 4; u++) hb2(2 * Y + 1 + u, n0[u], n1[u]);
#pragma unroll
            for (int u = 0; u < 2; u++) {
                if (Y + u < Y1) {
                    // blurred rows 2(Y+u), 2(Y+u)+1 at the two columns; the 2x2 mean in k_fb_resize's order
                    const float p0 = n0[2 * u], p1 = n1[2 * u], q0 = n0[2 * u + 1], q1 = n1[2 * u + 1];
                    float b00 = a00 * k1; b00 += (am0 + p0) * k2;
                    float b01 = a01 * k1; b01 += (am1 + p1) * k2;
                    float b10 = p0 * k1;  b10 += (a00 + q0) * k2;
                    float b11 = p1 * k1;  b11 += (a01 + q1) * k2;
                    o[(int64_t)(Y + u) * lw] = (b00 + b01 + b10 + b11) * 0.25f;
                    am0 = p0; am1 = p1; a00 = q0; a01 = q1;
                }
            }
        }
    }
}

// The level image of every plane in one launch (see k_fb_level).  T == nullptr: the finest level (lh x lw = h x w, no
// resize).  Returns false when the level's patch does not fit the LDS budget (caller falls back to the three-kernel path).
bool launch_fb_level(hipStream_t st, const uint8_t *gray, int pitch, int64_t plane_stride, int planes, int h, int w,
                     const fb_taps &K, const fb_resize_tabs *T, float *out, int lh, int lw)
{
    const int r = K.ksize >> 1;
    if (K.ksize == 3 && (!T || (T->mode == 1 && w == 2 * lw && h == 2 * lh))) {
        // strips of >= 32 level rows, ~4096 workgroups per launch (the kernel holds 8 per CU and each is a short chain)
        const int nbx = (lw + 255) / 256;
        int ns = (int)((4096 + (long long)nbx * planes - 1) / ((long long)nbx * planes));
        const int cap = lh / 32 < 1 ? 1 : lh / 32;
        ns = ns < 1 ? 1 : (ns > cap ? cap : ns);
        int QS = (lh + ns - 1) / ns;
        QS = (QS + 3) & ~3; // whole batches of rows in flight
        ns = (lh + QS - 1) / QS;
        const dim3 grid(nbx, ns, planes);
        if (!T)
            hipLaunchKernelGGL(k_fb_level3<1>, grid, dim3(256), 0, st, gray, pitch, plane_stride, h, w, K.k[1], K.k[2], out, lh, lw, QS);
        else
            hipLaunchKernelGGL(k_fb_level3<2>, grid, dim3(256), 0, st, gray, pitch, plane_stride, h, w, K.k[1], K.k[2], out, lh, lw, QS);
        return true;
    }
    fb_level_args A;
    A.gray = gray; A.pitch = pitch; A.plane_stride = plane_stride; A.h = h; A.w = w;
    A.out = out; A.lh = lh; A.lw = lw;
    A.sx = A.sy = nullptr; A.xa = A.yb = nullptr; A.mode = 0;
    // r = 1 (the two finest levels of a 0.5 pyramid): 62 samples per tile side, so the patch is 64 x 64 - one pass of the
    // column loop and two of the row batches instead of two and three
    const int ts = r <= 1 ? 62 : 64;
    if (!T) {
        A.cap_x = 64 + 2 * r; A.cap_y = 64 + 2 * r;
        A.tsx = A.tsy = ts;
        const size_t lds = fb_level_lds(A.cap_x, A.cap_y, 64);
        if (lds > 64 * 1024) return false;
        hipLaunchKernelGGL((k_fb_level<1, 64>), dim3((lw + ts - 1) / ts, (lh + ts - 1) / ts, planes), dim3(256), lds, st, A, K);
        return true;
    }
    if (!T->sx || !T->sy) return false;
    A.sx = T->sx; A.sy = T->sy; A.xa = T->xa; A.yb = T->yb; A.mode = T->mode;
    A.cap_x = T->span_x32 + 2 * r;
    // few taps: tall tiles (32 level rows); many taps (coarse levels): 8 level rows, the patch is 2r rows taller than the tile
    const bool tall = r <= 2;
    A.cap_y = (tall ? T->span_y32 : T->span_y8) + 2 * r;
    const size_t lds = fb_level_lds(A.cap_x, A.cap_y, tall ? 64 : 16);
    if (lds > 64 * 1024) return false;
    A.tsx = ts;
    A.tsy = tall ? ts : 16;
    const int ox = A.tsx / 2, oy = A.tsy / 2; // level pixels per tile
    if (tall)
        hipLaunchKernelGGL((k_fb_level<2, 64>), dim3((lw + ox - 1) / ox, (lh + oy - 1) / oy, planes), dim3(256), lds, st, A, K);
    else
        hipLaunchKernelGGL((k_fb_level<2, 16>), dim3((lw + ox - 1) / ox, (lh + oy - 1) / oy, planes), dim3(256), lds, st, A, K);
    return true;
}

void launch_fb_resize(hipStream_t st, const float *src, int sh, int sw, int cn, float *dst, int dh, int dw, int images,
                      const fb_resize_tabs &T, float mul, bool apply_mul)
{
    if (cn == 2 && T.mode == 0 && apply_mul && dw >= sw && dh >= sh) { // the pyramid's flow upsample
        hipLaunchKernelGGL(k_fb_upflow, dim3((dw + 255) / 256, (dh + UF_ROWS - 1) / UF_ROWS, images), dim3(256), 0, st, src, sh, sw, dst,
                           dh, dw, T.xofs, T.xa, T.yofs, T.yb, mul);
        return;
    }
    dim3 grid((dw + 255) / 256, (dh + FB_RB - 1) / FB_RB, images);
    if (cn == 1)
        hipLaunchKernelGGL(k_fb_resize<1>, grid, dim3(256), 0, st, src, sh, sw, dst, dh, dw, T.xofs, T.xa, T.yofs, T.yb,
                           T.mode, mul, (int)apply_mul);
    else
        hipLaunchKernelGGL(k_fb_resize<2>, grid, dim3(256), 0, st, src, sh, sw, dst, dh, dw, T.xofs, T.xa, T.yofs, T.yb,
                           T.mode, mul, (int)apply_mul);
}

void launch_fb_polyexp(hipStream_t st, const float *in, int planes, int h, int w, const fb_poly &C, float *out)
{
#if FB_POLY_MARCH
    // strips so that a launch has ~8 workgroups per CU, at least 32 rows each
    const int ncb = (w + PM_OUT - 1) / PM_OUT;
    int ns = (int)((2048 + (long long)ncb * planes - 1) / ((long long)ncb * planes));
    const int cap = h / 32 < 1 ? 1 : h / 32;
    ns = ns < 1 ? 1 : (ns > cap ? cap : ns);
    const int QS = (h + ns - 1) / ns;
    ns = (h + QS - 1) / QS;
    hipLaunchKernelGGL(k_fb_polyexp_march, dim3(ncb * ns, 1, planes), dim3(PM_NT), 0, st, in, h, w, C, out, ncb, QS);
#else
    dim3 grid((w + PE_TX - 1) / PE_TX, (h + PE_TY * PE_TILES - 1) / (PE_TY * PE_TILES), planes);
    hipLaunchKernelGGL(k_fb_polyexp, grid, dim3(256), 0, st, in, h, w, C, out);
#endif
}

#ifdef VQA_AB_VARIANTS
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

#endif

// One fused iteration: flow_out = solve(box15x15(products(R, flow))).  flow == nullptr: zero flow (coarsest level).
// mag_partials (fb_iter_blocks(..) doubles per pair, or null): also leave the partial sums of |flow_out| there.
// flow_out must not alias the input: neighbours read it while it is written.  (Upsampling the coarser field inside the
// first iteration, as rounds 2-3 did in k_fb_update<1>, put eight dependent loads per row on the march's critical path and
// cost more than writing the upsampled field once with k_fb_resize<2>: LAB_NOTES.md L6.)
// workgroups per pair of launch_fb_iter at this geometry (= partials per pair of its magnitude sums)
// most strips a frame is cut into: strips are at least 32 rows (and at least one restart period)
static int fb_iter_strip_cap(int h)
{
    const int m = FI_RESTART > 32 ? FI_RESTART : 32;
    return h / m < 1 ? 1 : (h / m > 64 ? 64 : h / m);
}

static void fb_iter_geometry(int pairs, int h, int w, int &ncb, int &ns, int &QS)
{
    ncb = (w + FI_OUT - 1) / FI_OUT;
    // Strips.  A workgroup's march is a chain of dependent steps, so a launch costs (residency rounds) x (rows a workgroup
    // marches): the chip holds 768 of these workgroups at once (3 per CU: 160 VGPRs, 48 KB LDS).  Take the strip count
    // that minimises rounds x (strip rows + 16 priming rows); strips are >= 32 rows and whole multiples of FI_RESTART rows, the rows where
    // the column sums restart anyway - so the choice (which follows the number of pairs) changes speed, never a bit.
    ns = 1;
    QS = (h + FI_RESTART - 1) / FI_RESTART * FI_RESTART;
    const int cap = fb_iter_strip_cap(h);
    long long best = -1;
    for (int n = 1; n <= cap; n++) {
        const int qs = ((h + n - 1) / n + FI_RESTART - 1) / FI_RESTART * FI_RESTART, ne = (h + qs - 1) / qs;
        constexpr int RES = 768 * 256 / FI_NT; // workgroups resident at once (3 waves per SIMD)
        const long long blocks = (long long)ncb * ne * pairs, cost = ((blocks + RES - 1) / RES) * (qs + 16);
        if (best < 0 || cost < best) { best = cost; ns = ne; QS = qs; }
    }
}

// upper bound of fb_iter_blocks over every pair count (sizes the partials buffer)
int fb_iter_max_blocks(int h, int w)
{
    return (w + FI_OUT - 1) / FI_OUT * (fb_iter_strip_cap(h) + 1);
}

int fb_iter_blocks(int pairs, int h, int w)
{
    int ncb, ns, QS;
    fb_iter_geometry(pairs, h, w, ncb, ns, QS);
    return ncb * ns;
}

void launch_fb_iter(hipStream_t st, const float *R, const float *flow, int pairs, int h, int w, float *flow_out, double *mag_partials)
{
    int ncb, ns, QS;
    fb_iter_geometry(pairs, h, w, ncb, ns, QS);
    dim3 grid(ncb * ns, pairs);
    if (flow && mag_partials)
        hipLaunchKernelGGL((k_fb_iter<0, true>), grid, dim3(FI_NT), 0, st, R, flow, h, w, flow_out, ncb, QS, mag_partials);
    else if (flow)
        hipLaunchKernelGGL((k_fb_iter<0, false>), grid, dim3(FI_NT), 0, st, R, flow, h, w, flow_out, ncb, QS, nullptr);
    else if (mag_partials)
        hipLaunchKernelGGL((k_fb_iter<2, true>), grid, dim3(FI_NT), 0, st, R, nullptr, h, w, flow_out, ncb, QS, mag_partials);
    else
        hipLaunchKernelGGL((k_fb_iter<2, false>), grid, dim3(FI_NT), 0, st, R, nullptr, h, w, flow_out, ncb, QS, nullptr);
}

// mean |flow| from the partial sums the last iteration left (nblk = fb_iter_blocks of that launch per pair)
void launch_fb_mag_finalize(hipStream_t st, const double *partials, int nblk, int pairs, int h, int w, bool first_valid,
                            vqa_frame_metrics *res)
{
    hipLaunchKernelGGL(k_fb_mag_finalize<true>, dim3((pairs + 63) / 64), dim3(64), 0, st, partials, nblk, pairs,
                       1.0 / ((double)h * (double)w), (int)first_valid, res);
}

#ifdef VQA_AB_VARIANTS
int fb_mag_blocks() { return FB_MAG_BLOCKS; }

void launch_fb_mag(hipStream_t st, const float *flow, int pairs, int h, int w, double *partials, bool first_valid,
                   vqa_frame_metrics *res)
{
    hipLaunchKernelGGL(k_fb_mag, dim3(FB_MAG_BLOCKS, pairs), dim3(256), 0, st, flow, (int64_t)h * w, partials);
    hipLaunchKernelGGL(k_fb_mag_finalize<false>, dim3((pairs + 63) / 64), dim3(64), 0, st, partials, FB_MAG_BLOCKS, pairs,
                       1.0 / ((double)h * (double)w), (int)first_valid, res);
}
#endif

} // namespace vqa
