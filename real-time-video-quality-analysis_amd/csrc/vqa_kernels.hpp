// vqa_kernels.hpp — host-side launchers of the gfx950 kernels (internal).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vqa.h"

// Build flavours (csrc/Makefile):
//   default            the product: one kernel per stage, no environment variable selects an arithmetic path.
//   -DVQA_AB_VARIANTS  (make lab) additionally compiles the superseded kernels of earlier rounds and their VQA_*_VARIANT /
//                      tuning selectors, for re-measurement (LAB_NOTES.md); results stay parity-tested.
//   -DVQA_TEST_SEAMS   (make lab) additionally compiles the fault-injection / stand-in hooks the tests use:
//                      VQA_COMM_FAKE_RCCL, VQA_HYST_MAX_ROUNDS, VQA_HYST_RESCUE_MAX_ROUNDS, VQA_FAIL_ENSURE_AT, VQA_FB_CHUNK_BYTES.
#ifdef VQA_AB_VARIANTS
#include <cstdlib>
#endif

namespace vqa {

// value of an A/B selector: the environment in the lab build, the shipped default otherwise (callers cache it)
inline int ab_knob(const char *name, int dflt)
{
#ifdef VQA_AB_VARIANTS
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
#else
    (void)name;
    return dflt;
#endif
}

// k_gray_hist.hip
void launch_bgr2gray_hist(hipStream_t st, const uint8_t *bgr, int n, int h, int w, int64_t frame_stride,
                          int64_t row_stride, uint8_t *gray, int gp, int64_t plane_stride, vqa_frame_metrics *res,
                          bool gray_hist, bool color_hist, bool sum2);
void launch_resize_planes(hipStream_t st, const uint8_t *bgr, int n, int h, int w, int64_t frame_stride,
                          int64_t row_stride, int rw, int rh, const int32_t *xofs, const int32_t *xa,
                          const int32_t *yofs, const int32_t *yb, int mode, uint8_t *planeA, uint8_t *planeB, int pp,
                          int64_t plane_stride, vqa_frame_metrics *res, bool gray_hist, bool color_hist, bool sum2);

// k_orb.hip
void launch_orb64(hipStream_t st, const uint8_t *bgr, int n, int h, int w, int64_t frame_stride, int64_t row_stride,
                  const int32_t *xofs, const int32_t *xa, const int32_t *yofs, const int32_t *yb, int mode,
                  vqa_frame_metrics *res);

// k_dct8.hip
int dct8_blocks_per_frame(int h, int w);
// wave slots of the CURRENT device for the marching kernel (CUs x resident workgroups x 4): queried once per ctx in
// vqa_create and handed to every launch - no process-wide cache of a per-device quantity
int dct8_wave_slots();
void launch_dct8(hipStream_t st, const uint8_t *planes, int pitch, int64_t plane_stride, int n, int h, int w,
                 bool energy, bool temporal, bool first_has_prev, double *partials, vqa_frame_metrics *res, int wave_slots);

// k_dct_full.hip
void launch_dct_full(hipStream_t st, const uint8_t *planes, int pitch, int64_t plane_stride, int n, int h, int w,
                     const float *cw, const float *ch, float *scratch, double *pe, double *pt, bool energy,
                     bool temporal, bool first_has_prev, vqa_frame_metrics *res);

void launch_dct_full_finalize(hipStream_t st, const double *pe, const double *pt, int tiles, int n, vqa_frame_metrics *res,
                              bool energy, bool temporal, bool first_has_prev);

// k_dct_fft.hip: the same full-frame metrics through FFT-based row / column passes (lengths that factor into 2, 3, 5)
constexpr int DCT_FFT_MAX_PASSES = 12;
struct dct_fft_plan {            // one per transform length
    int n, npass;
    int radix[DCT_FFT_MAX_PASSES];
    int m[DCT_FFT_MAX_PASSES], tstep[DCT_FFT_MAX_PASSES];   // per pass: n / radix, n / (Ns radix)   (Ns = product of the radices before it)
    uint32_t ns_magic[DCT_FFT_MAX_PASSES];                 // per pass: ceil(2^32 / Ns): j / Ns = mulhi(j, magic) for j Ns < 2^32 (Ns = 1: 0, see k_dct_fft.hip)
    const float2 *tw;            // device: tw[m] = e^{-2 pi i m / n}, m < n
    const float2 *post;          // device: post[k] = s_k (cos, sin)(pi k / 2n), s_0 = sqrt(1/n), s_k = sqrt(2/n)
};
bool dct_fft_factor(int n, int radix[DCT_FFT_MAX_PASSES], int *npass);
bool dct_fft_supported(int h, int w); // the plane takes the FFT passes (else k_dct_full.hip's dense products)
int dct_fft_tiles(int h, int w); // partial sums per frame the column pass writes (sizes pe / pt)
void launch_dct_full_fft(hipStream_t st, const uint8_t *planes, int pitch, int64_t plane_stride, int n, int h, int w,
                         const dct_fft_plan &plan_w, const dct_fft_plan &plan_h, float *scratch, double *pe, double *pt,
                         bool energy, bool temporal, bool first_has_prev, vqa_frame_metrics *res);

// k_canny.hip
struct canny_geom {
    int tiles_x, tiles_y;
};
canny_geom canny_tiles(int h, int w);
void launch_canny_nms(hipStream_t st, const uint8_t *gray, int pitch, int64_t plane_stride, int n, int h, int w,
                      int low, int high, unsigned long long *strong, unsigned long long *weak,
                      vqa_frame_metrics *res);
// hysteresis on the strong/weak bit-planes: round 0 visits every 64x64 tile, later rounds the
// tiles a neighbour enqueued (compact list + dedup flags); *out_count must be 0 at launch.
unsigned canny_hyst_tiles(int n, int h, int w);
constexpr int CANNY_HYST_SEGMENTS = 16; // work-list segments per frame (k_canny.hip: HSEG)
// stats != 0: every tile visit adds its relaxation steps to res[f].hyst_steps (VQA_OPT_HYST_STATS; one atomic per visit)
void launch_canny_hyst_all(hipStream_t st, unsigned long long *strong, const unsigned long long *weak, int n, int h,
                           int w, unsigned *queued, unsigned *out_list, unsigned *out_count, vqa_frame_metrics *res,
                           int stats);
void launch_canny_hyst_list(hipStream_t st, unsigned long long *strong, const unsigned long long *weak, int n, int h,
                            int w, unsigned *in_queued, const unsigned *in_list, const unsigned *in_count,
                            unsigned *out_queued, unsigned *out_list, unsigned *out_count, unsigned *zero_count,
                            vqa_frame_metrics *res, int stats);
void launch_canny_hyst_tail(hipStream_t st, unsigned long long *strong, const unsigned long long *weak, int n, int h,
                            int w, unsigned *list0, unsigned *cnt0, unsigned *q0, unsigned *list1, unsigned *cnt1,
                            unsigned *q1, int first_in, vqa_frame_metrics *res, int stats, int max_rounds, int rescue_max_rounds);
constexpr int CANNY_HYST_MAX_ROUNDS = 1 << 16; // the tail's drain bound (a 64x64-tile fixpoint over any frame ends far below); a frame
                                               // that hits it is finished by the rescue pass under the proof's bound (k_canny.hip)
void launch_canny_finish(hipStream_t st, const unsigned long long *strong, int n, int h, int w, vqa_frame_metrics *res);

// k_sad.hip
void launch_block_sad(hipStream_t st, const uint8_t *planes, int pitch, int64_t plane_stride, int n, int h, int w,
                      int range, bool first_has_prev, vqa_frame_metrics *res);

// k_farneback.hip
struct fb_taps {       // one Gaussian blur kernel (getGaussianKernel, CV_32F), ksize <= 31
    float k[32];
    int ksize;
};
struct fb_poly {       // FarnebackPrepareGaussian(n = 5, sigma = 1.2)
    float g[11], xg[11], xxg[11];
    double ig11, ig03, ig33, ig55;
};
struct fb_resize_tabs { // cv2.resize INTER_LINEAR tables for float data (device pointers)
    int32_t *xofs = nullptr, *yofs = nullptr;
    float *xa = nullptr, *yb = nullptr;
    int mode = 0;      // 1 = exact 2x decimation
    // the source columns / rows a bilinear downscale actually samples (sorted, unique); null = all of them
    int32_t *cols = nullptr, *rows = nullptr;
    int nc = 0, nr = 0;
    // the fused level kernel's view of a downscale: the two source columns / rows each level column / row samples
    // (clamps applied; 2 * dsize entries) and the widest extent of them over any 32 consecutive columns, 8 and 32 rows
    int32_t *sx = nullptr, *sy = nullptr;
    int span_x32 = 0, span_y8 = 0, span_y32 = 0;
};
// blurred values are produced only at the columns / rows listed (all when null): the level's resize reads nothing else
void launch_fb_blur(hipStream_t st, const uint8_t *gray, int pitch, int64_t plane_stride, int planes, int h, int w,
                    const fb_taps &T, const int32_t *cols, int nc, const int32_t *rows, int nr, float *tmp, float *out);
// Gaussian blur of the u8 planes + resize to the level in one kernel; T = nullptr: finest level (no resize).  false: the
// level does not fit its LDS budget - use launch_fb_blur (+ launch_fb_resize)
bool launch_fb_level(hipStream_t st, const uint8_t *gray, int pitch, int64_t plane_stride, int planes, int h, int w,
                     const fb_taps &K, const fb_resize_tabs *T, float *out, int lh, int lw);
void launch_fb_resize(hipStream_t st, const float *src, int sh, int sw, int cn, float *dst, int dh, int dw, int images,
                      const fb_resize_tabs &T, float mul, bool apply_mul);
void launch_fb_polyexp(hipStream_t st, const float *in, int planes, int h, int w, const fb_poly &C, float *out);
// one flow iteration (products + 15x15 box sums + solve, fused; the products never reach HBM).  flow == nullptr: zero
// flow (coarsest level); flow_out must not alias the input
void launch_fb_iter(hipStream_t st, const float *R, const float *flow, int pairs, int h, int w, float *flow_out,
                    double *mag_partials = nullptr);
// workgroups (= magnitude partials) per pair of that launch, and the mean |flow| from them
int fb_iter_blocks(int pairs, int h, int w);
int fb_iter_max_blocks(int h, int w); // >= fb_iter_blocks for every pair count
void launch_fb_mag_finalize(hipStream_t st, const double *partials, int nblk, int pairs, int h, int w, bool first_valid,
                            vqa_frame_metrics *res);
#ifdef VQA_AB_VARIANTS // the two-kernel form of rounds 2-3 (VQA_FB_VARIANT=1 in the lab build)
void launch_fb_update(hipStream_t st, const float *R, const float *flow, int pairs, int h, int w, float *M);
void launch_fb_update_first(hipStream_t st, const float *R, const float *coarse, int ch, int cw, const fb_resize_tabs &T,
                            float mul, int pairs, int h, int w, float *M);
void launch_fb_blur_solve(hipStream_t st, const float *M, int pairs, int h, int w, float *flow);
#endif
#ifdef VQA_AB_VARIANTS // the separate magnitude pass of rounds 2-3 (with the two-kernel iteration)
int fb_mag_blocks();
void launch_fb_mag(hipStream_t st, const float *flow, int pairs, int h, int w, double *partials, bool first_valid,
                   vqa_frame_metrics *res);
#endif

// k_quality.hip
int ssim_gauss_blocks(int h, int w);
void launch_quality_gauss(hipStream_t st, const uint8_t *ref, const uint8_t *dist, int n, int64_t ref_frame_stride,
                          int64_t dist_frame_stride, const vqa_plane_desc *planes, const int *idx, int count,
                          int n_planes, double *partials, int64_t partial_plane_stride, vqa_plane_metrics *res);
void launch_quality_ffmpeg(hipStream_t st, const uint8_t *ref, const uint8_t *dist, int n, int64_t ref_frame_stride,
                           int64_t dist_frame_stride, const vqa_plane_desc *planes, const int *idx, int count,
                           int n_planes, double *partials, int64_t partial_plane_stride, vqa_plane_metrics *res);
int ssim_ffmpeg_blocks(int h, int w);

} // namespace vqa
