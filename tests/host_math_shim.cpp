// Compiles the kernels' per-pixel arithmetic (csrc/vqa_math.hpp) for the HOST so the exact
// device formulas can be unit-tested without a GPU.  Test scaffolding, not product code.
#include "../real-time-video-quality-analysis_amd/csrc/vqa_math.hpp"

extern "C" {
void shim_dct8x8(const float *in, float *out)
{
    float v[64];
    for (int i = 0; i < 64; i++) v[i] = in[i];
    vqa::dct8x8(v);
    for (int i = 0; i < 64; i++) out[i] = v[i];
}
unsigned shim_gray(unsigned b, unsigned g, unsigned r) { return vqa::bgr2gray_u8(b, g, r); }
unsigned shim_vcombine(int s0, int s1, int b0, int b1) { return vqa::resize_vcombine(s0, s1, b0, b1); }
int shim_canny_classify(int m, int gx, int gy, const int *nb, int low, int high)
{
    return vqa::canny_classify(m, gx, gy, nb, low, high);
}
int shim_fast9_score(int v, const int *ring, int thr) { return vqa::fast9_score(v, ring, thr); }
float shim_ssim_moments(float mx, float my, float sq, float xy) { return vqa::ssim_from_moments(mx, my, sq, xy); }
float shim_ssim_ffmpeg_end1(int s1, int s2, int ss, int s12) { return vqa::ssim_ffmpeg_end1(s1, s2, ss, s12); }
}
