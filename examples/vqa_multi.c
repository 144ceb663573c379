/* vqa_multi.c — BASELINE configs[4] from plain C: one stream per GPU, one host thread per device, one scalar all-reduce.
 *
 * The reference spreads frames over a process pool (complexity_metrics.py:143-147) and has no communication layer;
 * SURVEY.md section 8e maps that to one context per device, zero cross-device frame traffic and ONE SUM all-reduce of
 * the pooled scalars (RCCL over xGMI).  This program is that host, without Python or torch:
 *
 *   1. vqa_device_count -> D (or the first argument, to use fewer devices)
 *   2. per device: a host thread creates its vqa_ctx, makes its own synthetic 3840x2160 stream resident in HBM
 *      (stream id = device ordinal), and runs R passes of the full complexity suite + PSNR/SSIM over the batch
 *   3. the main thread builds ONE communicator over all contexts (vqa_comm_create -> ncclCommInitAll), all-reduces
 *      {sum of per-frame SSIM, sum of DCT energy, frames} and prints per-device fps and the cross-stream means
 *
 *   gcc -O2 -pthread -Iinclude -o vqa_multi examples/vqa_multi.c -Lreal-time-video-quality-analysis_amd/csrc -lvqa_hip \
 *       -Wl,-rpath,$PWD/real-time-video-quality-analysis_amd/csrc -lm
 *   ./vqa_multi [devices [frames_per_batch [passes [height width]]]]
 *
 * Exit code 0 = every call succeeded on every device and the reduced frame count equals devices x frames.
 * With ONE device this runs everywhere the library runs (tests/test_abi.py does that on the GPU box); with several it
 * needs a multi-GPU node, which no run of this repository has had yet - it is the last mile that could be prepared
 * without the hardware.
 * Rehearsal: VQA_MULTI_REHEARSAL_DEVICE=<d> puts EVERY worker on device d (so `devices` may exceed the device count):
 * the host threads, contexts, streams and the communicator's bookkeeping then run for real on a one-GPU box.  RCCL
 * itself refuses two ranks on one device, so the rehearsal needs the LAB build of the library (-DVQA_TEST_SEAMS) with
 * VQA_COMM_FAKE_RCCL=1; against the shipped library vqa_comm_create reports the duplicate device and the program fails. */
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "vqa.h"

typedef struct worker {
    int device, stream_id, n, h, w, passes;
    vqa_ctx *ctx;
    int rc;
    char err[256];
    double seconds, ssim_sum, dct_sum;
    unsigned long long edges;
} worker;

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

#define WCHECK(call)                                                                                         \
    do {                                                                                                     \
        int rc_ = (call);                                                                                    \
        if (rc_ != VQA_OK) {                                                                                 \
            snprintf(wk->err, sizeof wk->err, "device %d: %s -> %s (%s)", wk->device, #call, vqa_strerror(rc_), \
                     wk->ctx ? vqa_last_hip_error(wk->ctx) : "");                                           \
            wk->rc = rc_;                                                                                    \
            goto done;                                                                                       \
        }                                                                                                    \
    } while (0)

/* a device's own stream: smooth texture that pans with the frame index, +-3 perturbation for the "encoded" copy */
static void synth_frame(uint8_t *ref, uint8_t *dist, int h, int w, int t, unsigned stream)
{
    unsigned s = 2463534242u ^ (stream * 2654435761u) ^ (unsigned)t * 40503u;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            s = s * 1664525u + 1013904223u;
            const int base = (((x + 2 * t) >> 2) * 5 + ((y + t) >> 2) * 3 + (int)(stream * 17u)) & 0xff;
            for (int c = 0; c < 3; c++) {
                int v = (base * (8 - c)) >> 3;
                v += (int)(s >> 30);
                v = v > 255 ? 255 : v;
                int d = v + (int)((s >> (8 + 3 * c)) % 7u) - 3;
                d = d < 0 ? 0 : (d > 255 ? 255 : d);
                ref[((size_t)y * w + x) * 3 + c] = (uint8_t)v;
                dist[((size_t)y * w + x) * 3 + c] = (uint8_t)d;
            }
        }
}

static void *run_device(void *arg)
{
    worker *wk = (worker *)arg;
    const int n = wk->n, h = wk->h, w = wk->w;
    const size_t fb = (size_t)h * w * 3;
    uint8_t *href = NULL, *hdist = NULL;
    void *dref = NULL, *ddist = NULL;
    vqa_frame_metrics *fm = NULL;
    vqa_plane_metrics *pm = NULL;
    wk->rc = VQA_OK;
    WCHECK(vqa_create(wk->device, &wk->ctx));
    WCHECK(vqa_alloc_pinned(wk->ctx, fb, (void **)&href));
    WCHECK(vqa_alloc_pinned(wk->ctx, fb, (void **)&hdist));
    WCHECK(vqa_alloc_device(wk->ctx, fb * (size_t)(n + 1), &dref));
    WCHECK(vqa_alloc_device(wk->ctx, fb * (size_t)(n + 1), &ddist));
    for (int t = 0; t <= n; t++) { /* the stream becomes resident in HBM frame by frame through one pinned buffer */
        synth_frame(href, hdist, h, w, t, (unsigned)wk->stream_id);
        WCHECK(vqa_copy_h2d(wk->ctx, (uint8_t *)dref + fb * (size_t)t, href, fb));
        WCHECK(vqa_copy_h2d(wk->ctx, (uint8_t *)ddist + fb * (size_t)t, hdist, fb));
        WCHECK(vqa_sync(wk->ctx));
    }
    fm = (vqa_frame_metrics *)calloc((size_t)n, sizeof *fm);
    pm = (vqa_plane_metrics *)calloc((size_t)n * 3, sizeof *pm);
    if (!fm || !pm) { wk->rc = VQA_ERR_OOM; snprintf(wk->err, sizeof wk->err, "device %d: host allocation", wk->device); goto done; }
    vqa_params p;
    vqa_default_params(&p);
    p.dct_mode = VQA_DCT_BLOCK8;
    vqa_plane_desc planes[3];
    for (int c = 0; c < 3; c++) {
        planes[c].width = w; planes[c].height = h; planes[c].offset = c; planes[c].row_stride = (int64_t)w * 3;
        planes[c].pixel_step = 3; planes[c].pad_ = 0;
    }
    const uint8_t *r1 = (const uint8_t *)dref + fb, *d1 = (const uint8_t *)ddist + fb;
    for (int pass = -1; pass < wk->passes; pass++) { /* pass -1 warms the context (allocations, first touch) */
        const double t0 = now_s();
        WCHECK(vqa_quality_submit(wk->ctx, r1, d1, VQA_MEM_DEVICE, n, (int64_t)fb, (int64_t)fb, planes, 3, VQA_SSIM_GAUSS));
        WCHECK(vqa_complexity_submit(wk->ctx, d1, (const uint8_t *)ddist, VQA_MEM_DEVICE, n, h, w, (int64_t)fb, (int64_t)w * 3,
                                     VQA_M_ALL, &p));
        WCHECK(vqa_quality_wait(wk->ctx, pm, n * 3));
        WCHECK(vqa_complexity_wait(wk->ctx, fm, n));
        if (pass >= 0) wk->seconds += now_s() - t0;
    }
    for (int i = 0; i < n; i++) {
        wk->ssim_sum += (pm[i * 3].ssim + pm[i * 3 + 1].ssim + pm[i * 3 + 2].ssim) / 3.0;
        wk->dct_sum += fm[i].dct_energy;
        wk->edges += fm[i].edge_count;
        if (fm[i].hyst_overflow || fabs(fm[i].dct_energy - (double)fm[i].sum_gray2) > 1e-4 * (double)fm[i].sum_gray2) {
            wk->rc = VQA_ERR_STATE;
            snprintf(wk->err, sizeof wk->err, "device %d frame %d: self-check failed (Parseval / hysteresis bound)", wk->device, i);
            goto done;
        }
    }
done:
    free(fm);
    free(pm);
    if (wk->ctx) {
        if (dref) vqa_free_device(wk->ctx, dref);
        if (ddist) vqa_free_device(wk->ctx, ddist);
        if (href) vqa_free_pinned(wk->ctx, href);
        if (hdist) vqa_free_pinned(wk->ctx, hdist);
    }
    return NULL;
}

int main(int argc, char **argv)
{
    int avail = 0;
    if (vqa_abi_version() != VQA_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 2; }
    if (vqa_device_count(&avail) != VQA_OK || avail <= 0) {
        fprintf(stderr, "vqa_device_count -> %s\n", vqa_strerror(VQA_ERR_NO_DEVICE));
        return 2;
    }
    const char *reh = getenv("VQA_MULTI_REHEARSAL_DEVICE");
    const int reh_dev = reh ? atoi(reh) : -1;
    if (reh && (reh_dev < 0 || reh_dev >= avail)) { fprintf(stderr, "VQA_MULTI_REHEARSAL_DEVICE out of range\n"); return 2; }
    int nd = argc > 1 ? atoi(argv[1]) : avail;
    const int n = argc > 2 ? atoi(argv[2]) : 16, passes = argc > 3 ? atoi(argv[3]) : 3;
    const int h = argc > 5 ? atoi(argv[4]) : 2160, w = argc > 5 ? atoi(argv[5]) : 3840;
    if (nd <= 0 || (!reh && nd > avail)) nd = avail;
    if (nd > 64 || n <= 0 || passes <= 0 || h < 16 || w < 16) { fprintf(stderr, "bad arguments\n"); return 2; }
    worker wk[64];
    pthread_t th[64];
    memset(wk, 0, sizeof wk);
    for (int d = 0; d < nd; d++) {
        wk[d].device = reh ? reh_dev : d; wk[d].stream_id = d; wk[d].n = n; wk[d].h = h; wk[d].w = w; wk[d].passes = passes;
        if (pthread_create(&th[d], NULL, run_device, &wk[d]) != 0) { fprintf(stderr, "pthread_create failed\n"); return 2; }
    }
    int bad = 0;
    for (int d = 0; d < nd; d++) {
        pthread_join(th[d], NULL);
        if (wk[d].rc != VQA_OK) { fprintf(stderr, "%s\n", wk[d].err); bad = 1; }
    }
    double total_fps = 0;
    if (!bad) {
        /* the path's one collective: row d of vals belongs to context d; afterwards every row holds the sum over devices */
        vqa_ctx *ctxs[64];
        double vals[64][3];
        for (int d = 0; d < nd; d++) {
            ctxs[d] = wk[d].ctx;
            vals[d][0] = wk[d].ssim_sum; vals[d][1] = wk[d].dct_sum; vals[d][2] = (double)n;
            const double fps = (double)n * passes / wk[d].seconds;
            total_fps += fps;
            printf("worker %d on device %d: %d x %dx%d frames x %d passes  %.1f frames/s  mean SSIM %.6f  edges/frame %.0f\n", d, wk[d].device, n, w, h, passes,
                   fps, wk[d].ssim_sum / n, (double)wk[d].edges / n);
        }
        vqa_comm *comm = NULL;
        const int rc = vqa_comm_create(ctxs, nd, &comm);
        if (rc == VQA_ERR_UNSUPPORTED) {
            puts("RCCL not installed: collective skipped");
        } else if (rc != VQA_OK) {
            fprintf(stderr, "vqa_comm_create -> %s (%s)\n", vqa_strerror(rc), vqa_comm_last_error(NULL));
            bad = 1;
        } else {
            const int arc = vqa_allreduce(comm, &vals[0][0], 3);
            if (arc != VQA_OK) { fprintf(stderr, "vqa_allreduce -> %s (%s)\n", vqa_strerror(arc), vqa_comm_last_error(comm)); bad = 1; }
            for (int d = 0; d < nd && !bad; d++) /* every rank holds the same sums */
                bad |= vals[d][2] != (double)n * nd || vals[d][0] != vals[0][0] || vals[d][1] != vals[0][1];
            if (!bad)
                printf("all-reduce over %d device(s): %.0f frames  mean SSIM %.6f  mean DCT energy %.6g  aggregate %.1f frames/s\n",
                       vqa_comm_size(comm), vals[0][2], vals[0][0] / vals[0][2], vals[0][1] / vals[0][2], total_fps);
            vqa_comm_destroy(comm);
        }
    }
    for (int d = 0; d < nd; d++)
        if (wk[d].ctx) vqa_destroy(wk[d].ctx);
    if (bad) return 1;
    puts("vqa_multi ok");
    return 0;
}
