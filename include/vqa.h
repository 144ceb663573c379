/*
 * vqa.h — C ABI of the MI355X-native per-frame video complexity / quality engine.
 *
 * This is the drop-in boundary for the ONE hot path of
 * zaki699/Real-Time-Video-Quality-Analysis (reference @ 2024-10-16):
 *
 *   complexity_metrics.py:246-310  calculate_average_scene_complexity
 *   complexity_metrics.py:128-148  process_in_batches        (the data-parallel map)
 *   complexity_metrics.py:313-579  process_*_frame           (the per-frame kernels)
 *   video_processing.py:270-297    run_ffmpeg_metrics        (PSNR + SSIM)
 *
 * The reference has no FFI of its own (it is Python over opencv_python and an
 * ffmpeg subprocess); the entry points below are what a ctypes binding for that
 * path binds (INTEGRATION.md shows the stub).  Plain pointers and sizes only —
 * no torch / numpy types.  Every function returns VQA_OK (0) or a negative
 * vqa_status; nothing throws.
 *
 * Threading: one vqa_ctx per device per host thread (a thread may own several: two that
 * measure alternate chunks and one that only copies, tied together by vqa_stream_wait, is
 * the pipeline this library is built for).  A ctx owns one HIP stream,
 * its scratch planes and a pinned result staging area; it is NOT thread-safe.
 * Different contexts may be used from different threads at the same time.
 * Buffers handed to a *_submit call must stay alive and unmodified until the
 * matching *_wait returns.
 * Process-wide state, all of it: (1) the HIP runtime; (2) the table of RCCL entry
 * points, filled once by the first vqa_comm_create* / vqa_comm_unique_id call
 * (a C++11 function-local static: concurrent first calls are safe); (3) immutable
 * function-local constants.  Everything mutable - scratch, cached tables, launch
 * geometry derived from the device (queried in vqa_create), options, error text -
 * is a field of the ctx.
 *
 * Memory: a ctx keeps what it has grown - scratch planes sized by the largest batch
 * seen (Farneback: up to ~13.5 GiB), result staging, and small per-geometry tables
 * (at most VQA_TABLE_CACHE_GEOMETRIES entries of each kind, least recently used
 * evicted) - so that a steady stream of batches allocates nothing.  vqa_trim gives
 * all of it back without destroying the ctx (the reference holds nothing between
 * calls: a process pool per call, complexity_metrics.py:143-147).
 *
 * There is NO CPU fallback: vqa_create fails with VQA_ERR_NO_DEVICE when no
 * gfx950 device is visible.
 *
 * Environment.  The shipped library reads exactly ONE environment variable, once per
 * vqa_create: VQA_OVERLAP (0 / 1) = the initial value of VQA_OPT_OVERLAP below (a
 * scheduling choice; results are identical either way).  No environment variable
 * selects a kernel or changes an arithmetic path.  (The Python binding additionally
 * honours VQA_LIB_PATH to load another build of this ABI, VQA_DEVICE / LOCAL_RANK for the default
 * device, VQA_MOTION for the default motion definition and VQA_ROCTX for roctx ranges.)  A separate
 * LAB build (`make -C csrc lab` -> lab/libvqa_hip_lab.so, vqa_build_flavour() != 0)
 * keeps superseded kernels and test seams behind VQA_*_VARIANT / VQA_COMM_FAKE_RCCL /
 * VQA_HYST_MAX_ROUNDS / VQA_HYST_RESCUE_MAX_ROUNDS / VQA_FAIL_ENSURE_AT / VQA_FB_CHUNK_BYTES; it is for re-measurement and fault
 * injection only and is never loaded by default.
 */
#ifndef VQA_H
#define VQA_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define VQA_API __attribute__((visibility("default")))
#else
#define VQA_API
#endif

#define VQA_ABI_VERSION 7
#define VQA_TABLE_CACHE_GEOMETRIES 16

typedef enum vqa_status {
    VQA_OK = 0,
    VQA_ERR_INVALID = -1,     /* bad argument (NULL, non-positive size, bad mask ...)   */
    VQA_ERR_NO_DEVICE = -2,   /* no HIP device / device index out of range              */
    VQA_ERR_HIP = -3,         /* a HIP runtime call failed (vqa_last_hip_error)         */
    VQA_ERR_OOM = -4,         /* device or pinned allocation failed                     */
    VQA_ERR_UNSUPPORTED = -5, /* valid request this build does not implement            */
    VQA_ERR_STATE = -6,       /* wait without submit, submit while one is pending ...   */
    VQA_ERR_INCOMPLETE = -7   /* a result would be inexact and is withheld: vqa_complexity_wait when a frame's Canny
                                 hysteresis did not reach its fixpoint (vqa_last_hip_error names the frame)            */
} vqa_status;

typedef struct vqa_ctx vqa_ctx;

/* where a frame pointer handed to *_submit lives */
typedef enum vqa_mem_kind {
    VQA_MEM_HOST = 0,   /* pageable or pinned host memory: the engine copies H2D on its stream */
    VQA_MEM_DEVICE = 1  /* device memory on the ctx's device (vqa_alloc_device, torch data_ptr) */
} vqa_mem_kind;

/* ---- metric selection (one bit per reference kernel) --------------------- */
#define VQA_M_GRAY_HIST     (1u << 0) /* process_histogram_frame        complexity_metrics.py:392-416 */
#define VQA_M_COLOR_HIST    (1u << 1) /* process_color_histogram_frame  complexity_metrics.py:418-475 */
#define VQA_M_DCT           (1u << 2) /* process_dct_frame              complexity_metrics.py:346-364 */
#define VQA_M_TEMPORAL_DCT  (1u << 3) /* process_temporal_dct_frame     complexity_metrics.py:543-579 */
#define VQA_M_EDGE          (1u << 4) /* process_edge_frame             complexity_metrics.py:477-504 */
#define VQA_M_MOTION        (1u << 5) /* process_frame_complexity       complexity_metrics.py:313-343
                                         (block-SAD substitute for Farneback)         */
#define VQA_M_ORB           (1u << 6) /* process_orb_frame_for_parallel complexity_metrics.py:367-389
                                         (always on the 64x64 thumbnail, as the reference hard-codes) */
#define VQA_M_ALL           0x7Fu

/* dct_mode */
#define VQA_DCT_AUTO   0 /* FULL when resize_w*resize_h <= 128*128, else BLOCK8            */
#define VQA_DCT_BLOCK8 1 /* 8x8 block DCT-II (north_star).  Energy == full-frame energy     */
#define VQA_DCT_FULL   2 /* one full-frame DCT-II, exactly what cv2.dct computes            */

/* motion_mode (what VQA_M_MOTION computes) */
#define VQA_MOTION_SAD       0 /* 16x16 block-SAD full search (north_star; default)          */
#define VQA_MOTION_FARNEBACK 1 /* cv2.calcOpticalFlowFarneback(.., 0.5, 3, 15, 3, 5, 1.2, 0)
                                  mean magnitude — what the reference computes (:340-343).
                                  Scratch on the device: 59 bytes per pixel and pair of a chunk,
                                  chunks of up to 12 GiB (64 pairs of 1080p = 7.8 GB) and never
                                  more than 80 % of what the device has free (halved again if the
                                  reservation still fails), kept by the ctx until vqa_trim /
                                  vqa_destroy                                                  */

/* ssim_mode */
#define VQA_SSIM_GAUSS  0 /* 11x11 Gaussian window, sigma 1.5 (north_star)                  */
#define VQA_SSIM_FFMPEG 1 /* FFmpeg vf_ssim: integer 8x8 window, stride 4 (what the
                             reference's subprocess really computes, video_processing.py:276) */

typedef struct vqa_params {
    int32_t resize_w, resize_h; /* cv2.resize target; 0 or == frame size -> native (copy)  */
    int32_t canny_low, canny_high; /* cv2.Canny thresholds; reference uses 100, 200 (:503)  */
    int32_t sad_range;          /* block-SAD search radius R, 0..7 (default 7)              */
    int32_t dct_mode;           /* VQA_DCT_*                                                */
    int32_t motion_mode;        /* VQA_MOTION_*                                             */
    int32_t reserved[9];        /* must be zero                                             */
} vqa_params;

/* Per-frame results of the complexity kernels.  Integer fields are exact
 * (bit-identical to the CPU reference); double fields carry float sums.    */
typedef struct vqa_frame_metrics {
    uint32_t hist_gray[256];     /* calcHist of gray(resize(frame))        (:404-412)       */
    uint32_t hist_bgr[3][256];   /* calcHist of resize(frame) channels B,G,R (:430,455-457) */
    uint64_t sum_gray2;          /* exact sum of squares of the DCT input plane (Parseval)  */
    double   dct_energy;         /* sum(dct(resize(gray(frame)))**2)       (:358-364)       */
    double   temporal_dct_l1;    /* sum|dct(prev) - dct(curr)|; 0 if no previous frame      */
    uint64_t sad_sum;            /* sum over blocks of the winning SAD                      */
    uint32_t sad_blocks;         /* number of 16x16 blocks measured; 0 if no previous frame */
    uint32_t mv_d2_hist[129];    /* blocks per winning dx^2+dy^2                            */
    uint32_t edge_count;         /* np.sum(cv2.Canny(gray,low,high) > 0)   (:503-504)       */
    uint32_t edge_strong;        /* pixels above `high` that survive NMS                    */
    uint32_t edge_weak;          /* NMS survivors in (low, high]                            */
    uint32_t has_prev;           /* 1 if a previous frame was available                     */
    uint32_t hyst_steps;         /* diagnostics only (scheduling-dependent): relaxation steps summed over tile visits;
                                    0 unless VQA_OPT_HYST_STATS is set on the ctx (it costs an atomic per visit) */
    uint32_t orb_keypoints;      /* len(ORB_create().detectAndCompute(gray64)[0])  (:385-389)  */
    uint32_t orb_response;       /* FAST score of that keypoint, 0 when there is none       */
    uint32_t hyst_overflow;      /* how the Canny hysteresis of this frame ended.  0: the tail reached the fixpoint (every frame
                                    seen so far).  2: the tail stopped at its round bound and the rescue pass completed the
                                    fixpoint on the device - edge_count is exact either way.  1 is never returned: a frame that
                                    neither pass finished makes vqa_complexity_wait fail with VQA_ERR_INCOMPLETE and zero the
                                    records - edge_count is a bit-exact count or there is no count (complexity_metrics.py:503-504) */
    double   flow_mag_mean;      /* VQA_MOTION_FARNEBACK: np.mean(|flow|) (:342-343); else 0.  Bar: 1e-4 relative against
                                    the CPU restatement of cv2.calcOpticalFlowFarneback, EXCEPT on frames where a border
                                    pixel's flow lies within float rounding of FarnebackUpdateMatrices' in-frame test
                                    (a discontinuity of the algorithm: either side is a valid evaluation; seen on
                                    35x31 / 129x34 noise frames, 4e-4 on the mean) - there 2e-3.
                                    Independent of how frames are batched (round 6): the running column sums of the flow
                                    iteration restart at every multiple of 16 rows - the only rows where a row strip may
                                    begin - and the magnitudes are summed in 2^-28 fixed point, so the strip count a launch
                                    picks from its number of pairs changes speed, never a bit (rounds 4-5: <= 1e-6 between
                                    batch sizes).  Every field of this record is independent of the batch. */
} vqa_frame_metrics;

/* One 8-bit plane inside a frame buffer (planar YUV plane, or one channel of
 * packed BGR with pixel_step = 3).                                          */
typedef struct vqa_plane_desc {
    int32_t width, height;
    int64_t offset;      /* bytes from the start of the frame                 */
    int64_t row_stride;  /* bytes between rows                                */
    int32_t pixel_step;  /* bytes between horizontally adjacent samples       */
    int32_t pad_;
} vqa_plane_desc;

typedef struct vqa_plane_metrics {
    uint64_t sse;   /* sum (ref - dist)^2 over the plane — FFmpeg psnr's per-component sum */
    double   ssim;  /* mean SSIM of the plane in the selected ssim_mode.  Independent of how frames are batched: the Gaussian
                       kernel sums the SSIM map in 2^-27 fixed point (integer sums are associative, so the strip geometry a
                       launch picks from its workgroup count cannot show; rounds 1-5 summed floats and differed by <= 1e-8
                       between batch sizes); vf_ssim's samples are summed in double, exactly for planes below ~2^28
                       samples.  The same frame pair gives the same bits in any batch; sse is exact */
} vqa_plane_metrics;

/* ---- lifecycle ------------------------------------------------------------ */
VQA_API int vqa_abi_version(void);
VQA_API const char *vqa_strerror(int status);
VQA_API int vqa_device_count(int *count);
VQA_API int vqa_create(int device, vqa_ctx **out);
VQA_API int vqa_destroy(vqa_ctx *ctx);
/* Gives back everything an idle ctx keeps between batches: every grow-only scratch buffer, the result staging and every
 * cached table (device and pinned host memory; the streams, events and options stay).  The next submit re-grows what it
 * needs.  VQA_ERR_STATE while a batch is pending (between a *_submit and its *_wait).                                    */
VQA_API int vqa_trim(vqa_ctx *ctx);
/* text of the last failing HIP call on this ctx ("" if none) */
VQA_API const char *vqa_last_hip_error(const vqa_ctx *ctx);
VQA_API void vqa_default_params(vqa_params *p);
/* 0 = the shipped library; bit 0 = built with the superseded A/B kernels, bit 1 = built with the test seams (lab build) */
VQA_API int vqa_build_flavour(void);

/* ---- per-ctx options (none of them changes a result) ------------------------ */
enum vqa_option {
    VQA_OPT_OVERLAP = 0,    /* 1 (default; initial value from VQA_OVERLAP if set): inside one complexity submit block-SAD,
                               the Canny chain and (VQA_DCT_FULL) the full-frame DCT run on side streams of the ctx next to
                               the 8x8 DCT / ORB / Farneback kernels and join before the results are copied (+3 % on the
                               full suite, +5 % with the reference's own definitions).  0: every kernel on the ctx stream
                               in program order - per-kernel event times (vqa_profile_*) are then free of overlap        */
    VQA_OPT_HYST_STATS = 1  /* 1: fill vqa_frame_metrics.hyst_steps (default 0)                                        */
};
/* VQA_ERR_STATE while a submit is pending on the ctx */
VQA_API int vqa_set_option(vqa_ctx *ctx, int option, int value);
VQA_API int vqa_get_option(const vqa_ctx *ctx, int option, int *value);

/* ---- memory --------------------------------------------------------------- */
VQA_API int vqa_alloc_pinned(vqa_ctx *ctx, size_t bytes, void **out);
VQA_API int vqa_free_pinned(vqa_ctx *ctx, void *p);
/* *out = 1 if the WHOLE range [p, p + bytes) is page-locked host memory HIP knows (vqa_alloc_pinned, hipHostMalloc,
 * hipHostRegister, a torch tensor with pin_memory=True; first and last byte are probed): a *_submit / vqa_copy_h2d from it is
 * a true asynchronous DMA.  0 for ordinary (pageable) host memory - the host side then stages through pinned buffers of
 * its own - for a range that leaves a registered region, and for device memory.                                          */
VQA_API int vqa_host_is_pinned(vqa_ctx *ctx, const void *p, size_t bytes, int *out);
VQA_API int vqa_alloc_device(vqa_ctx *ctx, size_t bytes, void **out);
VQA_API int vqa_free_device(vqa_ctx *ctx, void *p);
/* async on the ctx stream; host side should be pinned for true overlap */
VQA_API int vqa_copy_h2d(vqa_ctx *ctx, void *dst_device, const void *src_host, size_t bytes);
VQA_API int vqa_copy_d2h(vqa_ctx *ctx, void *dst_host, const void *src_device, size_t bytes);
VQA_API int vqa_sync(vqa_ctx *ctx);
/* Device-side ordering between two contexts of ONE device, no host wait: everything enqueued on `waiter`'s stream after this
 * call starts only after everything enqueued on `signaler`'s stream BEFORE this call has completed (hipEventRecord +
 * hipStreamWaitEvent).  What it is for: a host that feeds frames over PCIe keeps ONE context as its copy lane - all
 * vqa_copy_h2d calls go there, so uploads cross PCIe one after another, in chunk order - and lets the context that will
 * measure a chunk wait for it:  vqa_copy_h2d(copy_ctx, ...); vqa_stream_wait(work_ctx, copy_ctx); vqa_*_submit(work_ctx, ...).
 * The upload of chunk k+1 then runs under the kernels of chunk k (uploads enqueued on the work contexts' own streams all
 * start at once, share the link and finish together: the kernels wait for all of them - profiles/round6_api_trace_*.json).
 * Both contexts are the calling thread's.  VQA_ERR_INVALID for the same ctx twice or contexts of different devices.     */
VQA_API int vqa_stream_wait(vqa_ctx *waiter, vqa_ctx *signaler);
/* the ctx's hipStream_t, as an opaque pointer (for event timing by the caller) */
VQA_API void *vqa_stream(vqa_ctx *ctx);

/* ---- complexity kernels (replaces process_in_batches over process_*_frame) - */
/* frames: n packed BGR24 frames (what cv2.VideoCapture.read yields,
 * complexity_metrics.py:100), frame i at frames + i*frame_stride, rows of
 * 3*w bytes at row_stride (>= 3*w: padded rows, or a region of interest
 * inside larger frames; only the 3*w bytes of each row are ever read).
 * prev0: the frame preceding frames[0] (same geometry, row_stride and
 * mem_kind) or NULL; frame i's "previous" is frame i-1.
 * Asynchronous: returns once the work is enqueued.  Limits: h*w <= 2^28 pixels
 * (VQA_ERR_UNSUPPORTED beyond); n is bounded by memory only (batches above 32768
 * frames are enqueued as consecutive slices internally).
 * Failure: a submit that returns non-zero has left NOTHING in flight - whatever it had
 * already enqueued has completed on every stream of the ctx before the call returns,
 * no batch is pending, the caller's buffers are free again and the ctx stays usable
 * (the reference's convention: log, re-raise, nothing left running,
 * video_processing.py:295-297).  The same holds for vqa_quality_submit.          */
VQA_API int vqa_complexity_submit(vqa_ctx *ctx, const uint8_t *frames, const uint8_t *prev0, int mem_kind,
                          int n, int h, int w, int64_t frame_stride, int64_t row_stride,
                          uint32_t metric_mask, const vqa_params *params);
/* blocks until the submitted batch is done, then fills out[0..n) */
VQA_API int vqa_complexity_wait(vqa_ctx *ctx, vqa_frame_metrics *out, int n);

/* ---- quality kernels (replaces run_ffmpeg_metrics' psnr + ssim filters) ---- */
/* ref/dist: n frames each; every frame holds n_planes planes described by
 * planes[].  out of vqa_quality_wait: n*n_planes entries, frame-major.        */
VQA_API int vqa_quality_submit(vqa_ctx *ctx, const uint8_t *ref, const uint8_t *dist, int mem_kind, int n,
                       int64_t ref_frame_stride, int64_t dist_frame_stride,
                       const vqa_plane_desc *planes, int n_planes, int ssim_mode);
VQA_API int vqa_quality_wait(vqa_ctx *ctx, vqa_plane_metrics *out, int n_entries);

/* ---- per-kernel timing (HIP events on the ctx stream) ----------------------- */
enum vqa_kernel_id {
    VQA_K_GRAY_HIST = 0, /* BGR->gray + histograms, native resolution   */
    VQA_K_RESIZE = 1,    /* cv2.resize gather + gray + histograms       */
    VQA_K_DCT8 = 2,      /* 8x8 DCT energy + temporal L1                */
    VQA_K_DCT_FULL = 3,  /* full-frame DCT: FFT row + column passes, or (sizes that do not factor into 2,3,5) 4 dense products */
    VQA_K_CANNY_NMS = 4,
    VQA_K_CANNY_HYST = 5,
    VQA_K_SAD = 6,
    VQA_K_SSIM_GAUSS = 7,
    VQA_K_SSIM_FFMPEG = 8,
    VQA_K_ORB = 9,       /* FAST-9/16 + NMS on the 64x64 thumbnail's centre */
    VQA_K_FARNEBACK = 10, /* the whole Farneback pyramid (about 30 launches per chunk of pairs) */
    VQA_K_COUNT = 11
};
/* When enabled, every kernel launch made by a submit call is bracketed by a
 * hipEvent pair recorded on the ctx stream; the elapsed times are accumulated
 * per kernel id when the batch is waited for.                                 */
VQA_API int vqa_profile_enable(vqa_ctx *ctx, int on);
VQA_API int vqa_profile_read(vqa_ctx *ctx, int kernel_id, double *total_ms, int64_t *launches, int reset);
VQA_API const char *vqa_kernel_name(int kernel_id);

/* ---- the path's one collective: SUM all-reduce of pooled scalars (RCCL over xGMI) ---
 * The reference has no communication layer (its parallelism is a process pool over
 * frames, complexity_metrics.py:143-147); frame batches shard one stream per GPU and
 * only a handful of pooled float64 scalars ever cross devices.
 * RCCL is loaded on first use (dlopen librccl.so.1): VQA_ERR_UNSUPPORTED without it.   */
typedef struct vqa_comm vqa_comm;
#define VQA_COMM_ID_BYTES 128
/* single process, several devices: one ctx per device (ncclCommInitAll)                */
VQA_API int vqa_comm_create(vqa_ctx *const *ctxs, int n_ctx, vqa_comm **out);
/* one process per device: rank 0 makes an id, the host program ships its
 * VQA_COMM_ID_BYTES bytes to the other ranks by its own means, every rank joins        */
VQA_API int vqa_comm_unique_id(void *id, size_t id_bytes);
VQA_API int vqa_comm_create_rank(vqa_ctx *ctx, const void *id, size_t id_bytes, int n_ranks, int rank, vqa_comm **out);
VQA_API int vqa_comm_destroy(vqa_comm *comm);
VQA_API int vqa_comm_size(const vqa_comm *comm);           /* ranks in the communicator */
VQA_API const char *vqa_comm_last_error(const vqa_comm *comm); /* comm == NULL: why this thread's last creation failed */
/* Test seam, LAB BUILD ONLY (vqa_build_flavour() & 2).  There, with VQA_COMM_FAKE_RCCL=1 in the environment when the
 * first communicator is made, an in-library stand-in takes the place of the RCCL entry points (it checks the group
 * bracketing and sums on the host), which lets the single-process multi-context path run on a one-GPU box; this returns
 * the stand-in's call trace.  The shipped library has no stand-in: it always returns "".  Multi-device use over real
 * RCCL has not run on hardware yet (no multi-GPU box was available to the builder).                                  */
VQA_API const char *vqa_comm_debug_trace(void);
/* In place: vals is [local contexts][count] doubles, row i belongs to the i-th local
 * context (one row with vqa_comm_create_rank); on return every row holds the sum over
 * ALL ranks.  count <= 64.  Blocking; runs on the contexts' streams.                   */
VQA_API int vqa_allreduce(vqa_comm *comm, double *vals, int count);

/* ---- introspection for tests (device-side intermediates) ------------------ */
/* Copies intermediate planes of the LAST complexity batch to host memory:
 * which = 0: DCT input plane  (resize(gray(frame)))   [n][ph][pw]
 * which = 1: hist/edge plane  (gray(resize(frame)))   [n][ph][pw]
 * which = 2: Canny edge map, 0/255                    [n][ph][pw]
 * which = 3: full-resolution gray (motion input)      [n][h][w]              */
VQA_API int vqa_debug_read_plane(vqa_ctx *ctx, int which, int frame, uint8_t *dst, int dst_h, int dst_w);

#ifdef __cplusplus
}
#endif
#endif /* VQA_H */
