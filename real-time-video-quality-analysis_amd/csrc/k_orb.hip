// k_orb.hip — ORB keypoint count of the 64x64 thumbnail for gfx950.
//
// Reference function replaced: process_orb_frame_for_parallel, complexity_metrics.py:367-389
//   gray = cv2.cvtColor(cv2.resize(frame, (64, 64)), cv2.COLOR_BGR2GRAY)
//   len(cv2.ORB_create().detectAndCompute(gray, None)[0])
// With ORB's defaults (edgeThreshold 31, 8 levels at scale 1.2, fastThreshold 20) a 64x64 image keeps
// keypoints only at x, y in {31, 32} of pyramid level 0 (oracle/vqa_oracle.c, vqo_orb64_count, spells the
// size arithmetic out), so the count is the number of FAST-9/16 corners that survive the strict 3x3
// non-max suppression at those four pixels.  That needs FAST scores on the 4x4 pixels 30..33 and hence
// the thumbnail's 10x10 pixels 27..36 — 100 bilinear gathers per frame, whatever the source size.
//
// Mapping: one 128-thread workgroup per frame.  Threads 0..99 each produce one thumbnail pixel
// (cv2.resize fixed-point arithmetic on B, G, R, then BGR2GRAY, exactly as k_resize_planes' plane B),
// threads 0..15 score the 4x4 candidates, thread 0 applies NMS.  Latency-bound: ~1.2 kB read per frame.
#include "vqa_dev.hpp"
#include "vqa_kernels.hpp"
#include "vqa_math.hpp"

namespace vqa {

constexpr int ORB_P0 = 27, ORB_PN = 10; // thumbnail patch rows/cols 27..36

__global__ __launch_bounds__(128) void k_orb64(const uint8_t *__restrict__ bgr, int h, int w, int64_t frame_stride,
                                               int64_t row_stride, const int32_t *__restrict__ xofs,
                                               const int32_t *__restrict__ xa, const int32_t *__restrict__ yofs,
                                               const int32_t *__restrict__ yb, int mode, int fast_threshold,
                                               vqa_frame_metrics *__restrict__ res)
{
    __shared__ int patch[ORB_PN][ORB_PN];
    __shared__ int score[4][4];
    const int f = blockIdx.x, t = threadIdx.x;
    const uint8_t *src = bgr + (int64_t)f * frame_stride;
    if (t < ORB_PN * ORB_PN) {
        const int oy = ORB_P0 + t / ORB_PN, ox = ORB_P0 + t % ORB_PN;
        int sx0, sx1, sy0, sy1, a0 = 0, a1 = 0, b0 = 0, b1 = 0;
        if (mode == 1) { // exact 2x decimation: INTER_AREA fast path
            sx0 = 2 * ox; sx1 = sx0 + 1; sy0 = 2 * oy; sy1 = sy0 + 1;
        } else {
            sx0 = xofs[ox]; sx1 = min(sx0 + 1, w - 1);
            a0 = xa[2 * ox]; a1 = xa[2 * ox + 1];
            const int sy = yofs[oy];
            sy0 = min(max(sy, 0), h - 1); sy1 = min(max(sy + 1, 0), h - 1);
            b0 = yb[2 * oy]; b1 = yb[2 * oy + 1];
        }
        const uint8_t *r0 = src + (int64_t)sy0 * row_stride, *r1 = src + (int64_t)sy1 * row_stride;
        uint32_t rc[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const uint32_t c00 = r0[sx0 * 3 + k], c01 = r0[sx1 * 3 + k], c10 = r1[sx0 * 3 + k], c11 = r1[sx1 * 3 + k];
            rc[k] = mode == 1 ? (c00 + c01 + c10 + c11 + 2) >> 2
                              : resize_vcombine((int)(c00 * a0 + c01 * a1), (int)(c10 * a0 + c11 * a1), b0, b1);
        }
        patch[t / ORB_PN][t % ORB_PN] = (int)bgr2gray_u8(rc[0], rc[1], rc[2]);
    }
    __syncthreads();
    if (t < 16) {
        // candidate (30 + t%4, 30 + t/4) sits at patch[3 + t/4][3 + t%4]
        const int py = 3 + (t >> 2), px = 3 + (t & 3);
        const int dx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
        const int dy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};
        int ring[16];
#pragma unroll
        for (int k = 0; k < 16; k++) ring[k] = patch[py + dy[k]][px + dx[k]];
        score[t >> 2][t & 3] = fast9_score(patch[py][px], ring, fast_threshold);
    }
    __syncthreads();
    if (t == 0) {
        uint32_t n = 0, best = 0;
        for (int y = 1; y <= 2; y++)
            for (int x = 1; x <= 2; x++) {
                const int s = score[y][x];
                bool keep = s > 0;
                for (int v = -1; v <= 1; v++)
                    for (int u = -1; u <= 1; u++)
                        if (u || v) keep = keep && (s > score[y + v][x + u]);
                if (keep) { n++; best = max(best, (uint32_t)s); }
            }
        res[f].orb_keypoints = n;
        res[f].orb_response = best;
    }
}

void launch_orb64(hipStream_t st, const uint8_t *bgr, int n, int h, int w, int64_t frame_stride, int64_t row_stride,
                  const int32_t *xofs, const int32_t *xa, const int32_t *yofs, const int32_t *yb, int mode,
                  vqa_frame_metrics *res)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_orb64, dim3(n), dim3(128), 0, st, bgr, h, w, frame_stride, row_stride, xofs, xa, yofs, yb, mode,
                       20 /* cv2.ORB_create() default fastThreshold */, res);
}

} // namespace vqa
