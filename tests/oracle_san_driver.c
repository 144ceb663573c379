/* Sanitizer driver for the CPU oracle (tests/test_oracle_sanitizers.py): every vqo_* entry point on seeded inputs of
 * awkward geometry (1-pixel planes, ragged sizes, strided channel views), compiled together with oracle/vqa_oracle.c under
 * -fsanitize=address,undefined -fno-sanitize-recover.  GPU sanitizers are not available on the pool; this is the CPU
 * half the task statement allows.  Exit 0 and "SAN-OK" = no report.  (Test infrastructure, like the oracle itself.) */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stddef.h>

/* the oracle is compiled INTO this translation unit, so every call below is checked against the real prototypes */
#include "../oracle/vqa_oracle.c"

static unsigned rs = 12345u;
static unsigned rnd(void) { rs = rs * 1664525u + 1013904223u; return rs >> 8; }

/* exact-size heap buffers so that AddressSanitizer sees every overrun */
static uint8_t *plane(size_t n, int smooth)
{
    uint8_t *p = (uint8_t *)malloc(n ? n : 1);
    for (size_t i = 0; i < n; i++) p[i] = smooth ? (uint8_t)((i * 7 / 5 + (rnd() & 7)) & 0xff) : (uint8_t)(rnd() & 0xff);
    return p;
}

int main(void)
{
    static const int geo[][2] = {{1, 1}, {1, 17}, {9, 1}, {8, 8}, {11, 11}, {16, 16}, {17, 31}, {33, 47}, {64, 64}, {67, 131}, {96, 200}};
    double sink = 0;
    for (unsigned gi = 0; gi < sizeof geo / sizeof geo[0]; gi++) {
        const int h = geo[gi][0], w = geo[gi][1];
        const size_t P = (size_t)h * w;
        uint8_t *bgr0 = plane(P * 3, gi & 1), *bgr1 = plane(P * 3, gi & 1);
        uint8_t *g0 = (uint8_t *)malloc(P), *g1 = (uint8_t *)malloc(P);
        vqo_bgr2gray(bgr0, h, w, (ptrdiff_t)w * 3, g0, w);
        vqo_bgr2gray(bgr1, h, w, (ptrdiff_t)w * 3, g1, w);
        uint32_t hist[256];
        vqo_hist_u8(g0, P, 1, hist);
        for (int c = 0; c < 3; c++) { vqo_hist_u8(bgr0 + c, P, 3, hist); sink += hist[7]; }
        double d3[3];
        vqo_dct8x8(g0, g1, h, w, w, d3);
        sink += d3[0] + d3[1];
        if (P <= 64 * 64) { sink += vqo_dct_energy_full(g1, h, w) + vqo_temporal_dct_full(g0, g1, h, w); }
        int16_t *dx = (int16_t *)malloc(sizeof(int16_t) * P), *dy = (int16_t *)malloc(sizeof(int16_t) * P);
        int32_t *mag = (int32_t *)malloc(sizeof(int32_t) * P);
        vqo_sobel_l1(g0, h, w, w, dx, dy, mag);
        uint8_t *edges = (uint8_t *)malloc(P);
        long ns = 0, nw = 0;
        sink += (double)vqo_canny_count(g0, h, w, w, 100, 200, edges, &ns, &nw);
        sink += (double)vqo_canny_count(g1, h, w, w, 300, 20, NULL, &ns, &nw);
        for (int range = 0; range <= 7; range += 7) {
            uint64_t sad = 0;
            uint32_t mh[129];
            const int nb = (h / 16) * (w / 16);
            int8_t *mv = (int8_t *)malloc((size_t)(nb ? nb : 1) * 2);
            sink += vqo_block_sad(g0, g1, h, w, w, range, &sad, mh, mv) + (double)sad;
            free(mv);
        }
        for (int c = 0; c < 3; c++) sink += (double)vqo_sse_plane(bgr0 + c, (ptrdiff_t)w * 3, bgr1 + c, (ptrdiff_t)w * 3, h, w, 3);
        if (h >= 11 && w >= 11) sink += vqo_ssim_gauss(bgr0 + 1, (ptrdiff_t)w * 3, bgr1 + 1, (ptrdiff_t)w * 3, h, w, 3) + vqo_ssim_gauss(g0, w, g1, w, h, w, 1);
        if (h >= 8 && w >= 8) sink += vqo_ssim_ffmpeg(bgr0 + 2, (ptrdiff_t)w * 3, bgr1 + 2, (ptrdiff_t)w * 3, h, w, 3) + vqo_ssim_ffmpeg(g0, w, g1, w, h, w, 1);
        int32_t *score = (int32_t *)malloc(sizeof(int32_t) * P);
        uint8_t *keep = (uint8_t *)malloc(P);
        sink += (double)vqo_fast9(g0, h, w, w, 20, 1, score, keep) + (double)vqo_fast9(g1, h, w, w, 5, 0, score, keep);
        /* resize: down, up, and the exact 2x shortcut; 1 and 3 channels */
        static const int tg[][2] = {{1, 1}, {5, 3}, {64, 64}, {40, 24}};
        for (unsigned ti = 0; ti < 4; ti++) {
            const int dh = tg[ti][1], dw = tg[ti][0];
            uint8_t *d1 = (uint8_t *)malloc((size_t)dh * dw), *d3b = (uint8_t *)malloc((size_t)dh * dw * 3);
            sink += vqo_resize_linear(g0, h, w, 1, d1, dh, dw) + vqo_resize_linear(bgr0, h, w, 3, d3b, dh, dw) + d1[0] + d3b[0];
            if (dh == 64 && dw == 64) { int resp = 0; sink += vqo_orb64_count(d1, 64, &resp) + resp; }
            free(d1); free(d3b);
        }
        if (!(h & 1) && !(w & 1)) {
            uint8_t *d = (uint8_t *)malloc((size_t)(h / 2) * (w / 2) * 3);
            sink += vqo_resize_linear(bgr0, h, w, 3, d, h / 2, w / 2) + d[0];
            free(d);
        }
        if (h >= 4 && w >= 4 && P <= 96 * 200) {
            float *flow = (float *)malloc(sizeof(float) * P * 2);
            sink += vqo_farneback_mean_mag(g0, g1, h, w, w, flow) + vqo_farneback_mean_mag(g0, g1, h, w, w, NULL);
            free(flow);
        }
        free(bgr0); free(bgr1); free(g0); free(g1); free(dx); free(dy); free(mag); free(edges); free(score); free(keep);
    }
    printf("SAN-OK %g\n", sink);
    return 0;
}
