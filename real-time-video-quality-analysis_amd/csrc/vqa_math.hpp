// vqa_math.hpp — per-pixel arithmetic shared by the HIP kernels.
// Pure functions, no memory access: compiled for the device by hipcc and, with
// VQA_HD empty, for the host by tests/host_math_shim.cpp so the exact device
// arithmetic can be unit-tested on a machine without a GPU.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define VQA_HD __host__ __device__ __forceinline__
#else
#define VQA_HD inline
#endif

namespace vqa {

// cv2.cvtColor(BGR2GRAY) on uint8 (complexity_metrics.py:327,358,405,493,530):
// OpenCV 4.x 15-bit coefficients, CV_DESCALE rounding.
VQA_HD uint32_t bgr2gray_u8(uint32_t b, uint32_t g, uint32_t r)
{
    return (b * 3735u + g * 19235u + r * 9798u + 16384u) >> 15;
}

// cv2.resize INTER_LINEAR uint8 vertical combine of two horizontally
// interpolated rows (resize.cpp VResizeLinear<uchar,int,short,...>).
VQA_HD uint32_t resize_vcombine(int32_t s0, int32_t s1, int32_t b0, int32_t b1)
{
    return (uint32_t)((((b0 * (s0 >> 4)) >> 16) + ((b1 * (s1 >> 4)) >> 16) + 2) >> 2);
}

// Orthonormal 8-point DCT-II, in place, even/odd decomposition with the
// orthonormal scale folded into the constants (22 mul/fma + 12 add).
//   X_k = s_k * sum_n x_n cos((2n+1) k pi / 16),  s_0 = 1/sqrt(8), s_k = 1/2
template <int STRIDE>
VQA_HD void dct8(float *v)
{
    const float r8 = 0.35355339059327373f;  // 1/sqrt(8)
    const float c1 = 0.49039264020161522f;  // cos(1 pi/16)/2
    const float c2 = 0.46193976625564337f;  // cos(2 pi/16)/2
    const float c3 = 0.41573480615127262f;  // cos(3 pi/16)/2
    const float c5 = 0.27778511650980114f;  // cos(5 pi/16)/2
    const float c6 = 0.19134171618254492f;  // cos(6 pi/16)/2
    const float c7 = 0.09754516100806417f;  // cos(7 pi/16)/2
    const float x0 = v[0 * STRIDE], x1 = v[1 * STRIDE], x2 = v[2 * STRIDE], x3 = v[3 * STRIDE];
    const float x4 = v[4 * STRIDE], x5 = v[5 * STRIDE], x6 = v[6 * STRIDE], x7 = v[7 * STRIDE];
    const float s0 = x0 + x7, s1 = x1 + x6, s2 = x2 + x5, s3 = x3 + x4;
    const float d0 = x0 - x7, d1 = x1 - x6, d2 = x2 - x5, d3 = x3 - x4;
    const float t0 = s0 + s3, t1 = s1 + s2, t2 = s0 - s3, t3 = s1 - s2;
    v[0 * STRIDE] = (t0 + t1) * r8;
    v[4 * STRIDE] = (t0 - t1) * r8;
    v[2 * STRIDE] = c2 * t2 + c6 * t3;
    v[6 * STRIDE] = c6 * t2 - c2 * t3;
    v[1 * STRIDE] = c1 * d0 + c3 * d1 + c5 * d2 + c7 * d3;
    v[3 * STRIDE] = c3 * d0 - c7 * d1 - c1 * d2 - c5 * d3;
    v[5 * STRIDE] = c5 * d0 - c1 * d1 + c7 * d2 + c3 * d3;
    v[7 * STRIDE] = c7 * d0 - c5 * d1 + c3 * d2 - c1 * d3;
}

#if defined(__HIPCC__)
// The same butterflies on float2 values: two independent transforms ride in the two halves of every
// register pair, so each add/mul/fma below is one packed instruction (v_pk_add/mul/fma_f32).
typedef float vqa_f2 __attribute__((ext_vector_type(2)));
template <int STRIDE>
__device__ __forceinline__ void dct8_x2(vqa_f2 *v)
{
    auto K = [](float c) { return vqa_f2{c, c}; };
    auto fma2 = [](vqa_f2 a, vqa_f2 b, vqa_f2 c) { return __builtin_elementwise_fma(a, b, c); };
    const float r8 = 0.35355339059327373f, c1 = 0.49039264020161522f, c2 = 0.46193976625564337f,
                c3 = 0.41573480615127262f, c5 = 0.27778511650980114f, c6 = 0.19134171618254492f,
                c7 = 0.09754516100806417f;
    const vqa_f2 x0 = v[0 * STRIDE], x1 = v[1 * STRIDE], x2 = v[2 * STRIDE], x3 = v[3 * STRIDE];
    const vqa_f2 x4 = v[4 * STRIDE], x5 = v[5 * STRIDE], x6 = v[6 * STRIDE], x7 = v[7 * STRIDE];
    const vqa_f2 s0 = x0 + x7, s1 = x1 + x6, s2 = x2 + x5, s3 = x3 + x4;
    const vqa_f2 d0 = x0 - x7, d1 = x1 - x6, d2 = x2 - x5, d3 = x3 - x4;
    const vqa_f2 t0 = s0 + s3, t1 = s1 + s2, t2 = s0 - s3, t3 = s1 - s2;
    v[0 * STRIDE] = (t0 + t1) * K(r8);
    v[4 * STRIDE] = (t0 - t1) * K(r8);
    v[2 * STRIDE] = fma2(K(c2), t2, K(c6) * t3);
    v[6 * STRIDE] = fma2(K(c6), t2, K(-c2) * t3);
    v[1 * STRIDE] = fma2(K(c1), d0, fma2(K(c3), d1, fma2(K(c5), d2, K(c7) * d3)));
    v[3 * STRIDE] = fma2(K(c3), d0, fma2(K(-c7), d1, fma2(K(-c1), d2, K(-c5) * d3)));
    v[5 * STRIDE] = fma2(K(c5), d0, fma2(K(-c1), d1, fma2(K(c7), d2, K(c3) * d3)));
    v[7 * STRIDE] = fma2(K(c7), d0, fma2(K(-c5), d1, fma2(K(c3), d2, K(-c1) * d3)));
}
__device__ __forceinline__ void dct8x8_x2(vqa_f2 *v)
{
#pragma unroll
    for (int r = 0; r < 8; r++) dct8_x2<1>(v + 8 * r);
#pragma unroll
    for (int c = 0; c < 8; c++) dct8_x2<8>(v + c);
}
#endif

// 2-D 8x8 DCT of a row-major block held in 64 registers.
VQA_HD void dct8x8(float *v)
{
#pragma unroll
    for (int r = 0; r < 8; r++) dct8<1>(v + 8 * r);
#pragma unroll
    for (int c = 0; c < 8; c++) dct8<8>(v + c);
}

// cv2.Canny non-maximum suppression + double threshold for one pixel
// (OpenCV 4.x canny.cpp; reference call complexity_metrics.py:503).
//   m    : L1 magnitude at the pixel;  gx, gy: Sobel responses (int16 range)
//   nb[] : magnitudes of the 8 neighbours, row-major without the centre:
//          0 1 2
//          3 . 4
//          5 6 7        (0 outside the image)
// returns 0 = not an edge, 1 = weak candidate, 2 = strong edge
VQA_HD int canny_classify(int m, int gx, int gy, const int nb[8], int low, int high)
{
    // Deliberately BRANCHY: on real frames most pixels fail m > low, and a wave whose lanes all fail
    // skips the sector test outright.  A select-only (branchless) form of the same truth table measured
    // 1.5x (register kernel) to 2x (LDS kernel) slower on natural content.
    if (!(m > low)) return 0;
    const int ax = gx < 0 ? -gx : gx;
    const int ay = (gy < 0 ? -gy : gy) << 15;
    const int tg22x = ax * 13573;
    bool keep;
    if (ay < tg22x) {
        keep = (m > nb[3]) && (m >= nb[4]);
    } else {
        const int tg67x = tg22x + (ax << 16);
        if (ay > tg67x) {
            keep = (m > nb[1]) && (m >= nb[6]);
        } else {
            // (xs ^ ys) < 0 on int16 values: opposite signs (zero counts as positive)
            const bool opposite = ((gx ^ gy) < 0);
            keep = opposite ? ((m > nb[2]) && (m > nb[5])) : ((m > nb[0]) && (m > nb[7]));
        }
    }
    if (!keep) return 0;
    return m > high ? 2 : 1;
}

// FAST-9/16 corner score of one pixel (OpenCV 4.x fast.cpp / fast_score.cpp cornerScore<16>, as used by
// cv2.ORB_create().detectAndCompute, complexity_metrics.py:385-387).
//   v      : the centre pixel;  ring[k]: the 16 pixels of the radius-3 Bresenham circle in circular order
//   thr    : the FAST threshold (ORB's default fastThreshold = 20)
// The pixel is a corner iff some run of 9 contiguous ring pixels is entirely darker than v - thr or
// entirely brighter than v + thr (strict).  With A = max over the 16 runs of min(v - ring) and
// B = max over the runs of min(ring - v), that is max(A, B) > thr, and the score OpenCV attaches to the
// corner (the largest threshold at which it would still be one) is max(A, B) - 1.
// returns the score for a corner, 0 otherwise (a corner's score is >= thr >= 1 whenever thr >= 1)
VQA_HD int fast9_score(int v, const int ring[16], int thr)
{
    int best = -256;
#pragma unroll
    for (int s = 0; s < 16; s++) {
        int lo = 255, hi = 255; // min(v - ring), min(ring - v) over the run s .. s+8
#pragma unroll
        for (int j = 0; j < 9; j++) {
            const int d = v - ring[(s + j) & 15];
            lo = d < lo ? d : lo;
            hi = -d < hi ? -d : hi;
        }
        const int m = lo > hi ? lo : hi;
        best = m > best ? m : best;
    }
    return best > thr ? best - 1 : 0;
}

// SSIM index from the four window moments used by the Gaussian kernel:
//   mx, my = E[x], E[y];  sq = E[x^2 + y^2];  xy = E[x y]
VQA_HD float ssim_from_moments(float mx, float my, float sq, float xy)
{
    const float C1 = 6.5025f;  // (0.01*255)^2
    const float C2 = 58.5225f; // (0.03*255)^2
    const float mxy = mx * my;
    const float m2 = mx * mx + my * my;
    const float num = (2.f * mxy + C1) * (2.f * (xy - mxy) + C2);
    const float den = (m2 + C1) * ((sq - m2) + C2);
    return num / den;
}

// FFmpeg vf_ssim.c ssim_end1: integer moments of an 8x8 window -> float index.
VQA_HD float ssim_ffmpeg_end1(int s1, int s2, int ss, int s12)
{
    const int c1 = 416;    // (int)(.01*.01*255*255*64 + .5)
    const int c2 = 235963; // (int)(.03*.03*255*255*64*63 + .5)
    const int vars = ss * 64 - s1 * s1 - s2 * s2;
    const int covar = s12 * 64 - s1 * s2;
    return (float)(2 * s1 * s2 + c1) * (float)(2 * covar + c2) /
           ((float)(s1 * s1 + s2 * s2 + c1) * (float)(vars + c2));
}

} // namespace vqa
