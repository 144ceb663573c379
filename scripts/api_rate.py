"""Rate of the reference-shaped entry point process_video_and_extract_metrics (quality of all frames + complexity
of every interval-th, ONE pass) on a 1080p clip that lives in HBM, in pinned host memory and in pageable host memory."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
if os.environ.get("VQA_BIND_NUMA", "1") != "0":   # before any GPU call: this process, its copier threads and its pinned pages next to the GPU
    from rtvqa_amd.affinity import bind_numa
    print("cpu affinity:", bind_numa(int(os.environ.get("VQA_DEVICE", "0"))), flush=True)
from rtvqa_amd import complexity_metrics as cm, synth, video_processing as vp, stream

n = int(sys.argv[1]) if len(sys.argv) > 1 else 257
h, w = int(os.environ.get("API_H", "1080")), int(os.environ.get("API_W", "1920"))
interval = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rs = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (w, h)
eng = cm.get_engine()
t0 = time.perf_counter()
pin_r, pin_d = eng.alloc_pinned((n, h, w, 3)), eng.alloc_pinned((n, h, w, 3))
print("pinned alloc of %.2f GB: %.2f s" % (2 * pin_r.nbytes / 1e9, time.perf_counter() - t0), flush=True)
for a in range(0, n, 32):
    r = synth.s_natural(min(32, n - a), h, w, seed=1234, t0=a)
    pin_r[a:a + len(r)] = r
    pin_d[a:a + len(r)] = synth.distort(r, t0=a)
ref, dist = np.array(pin_r), np.array(pin_d)
dev_r, dev_d = eng.upload(ref), eng.upload(dist)
cfg = {"crf": 23, "resize_width": rs[0], "resize_height": rs[1], "frame_interval": interval, "batch_size": 100}
csv = os.path.join(tempfile.gettempdir(), "api_rate.csv")
for name, (r, d) in (("resident", (dev_r, dev_d)), ("host_pinned", (pin_r, pin_d)), ("host_pageable", (ref, dist))):
    t0 = time.perf_counter()
    vp.process_video_and_extract_metrics(r, d, cfg, csv_file=csv)
    warm = time.perf_counter() - t0
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        m = vp.process_video_and_extract_metrics(r, d, cfg, csv_file=csv)
        best = min(best, time.perf_counter() - t0)
    print("%-14s %8.0f frames/s (%d frames, best of 3: %.1f ms; first call %.1f ms)  PSNR %.2f SSIM %.4f"
          % (name, n / best, n, best * 1e3, warm * 1e3, m["PSNR"], m["SSIM"]), flush=True)
for lanes in (1, 2, 3, 4):
    stream.MAX_LANES = lanes
    cm.release_buffers()
    out = []
    for name, (r, d) in (("resident", (dev_r, dev_d)), ("host_pinned", (pin_r, pin_d)), ("host_pageable", (ref, dist))):
        vp.process_video_and_extract_metrics(r, d, cfg, csv_file=csv)
        best = 1e9
        for _ in range(4):
            t0 = time.perf_counter()
            vp.process_video_and_extract_metrics(r, d, cfg, csv_file=csv)
            best = min(best, time.perf_counter() - t0)
        out.append("%s %6.0f" % (name, n / best))
    print("lanes %d: %s frames/s" % (lanes, "  ".join(out)), flush=True)
stream.MAX_LANES = 2
SWEEPS = os.environ.get("API_SWEEPS", "1") != "0"
for mb in ((64, 128, 160, 256, 512, 1024) if SWEEPS else ()):
    stream.CHUNK_BYTES_MAX = stream.STAGED_CHUNK_BYTES_MAX = mb << 20
    cm.release_buffers()
    out = []
    for name, (r, d) in (("host_pinned", (pin_r, pin_d)), ("host_pageable", (ref, dist))):
        vp.process_video_and_extract_metrics(r, d, cfg, csv_file=csv)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            vp.process_video_and_extract_metrics(r, d, cfg, csv_file=csv)
            best = min(best, time.perf_counter() - t0)
        out.append("%s %6.0f" % (name, n / best))
    print("chunk cap %4d MiB (both streams): %s frames/s" % (mb, "  ".join(out)), flush=True)
stream.CHUNK_BYTES_MAX, stream.STAGED_CHUNK_BYTES_MAX = 256 << 20, 256 << 20
for th in ((1, 2, 4, 8, 12) if SWEEPS else ()):
    stream.STAGE_THREADS = th
    cm.release_buffers()
    vp.process_video_and_extract_metrics(ref, dist, cfg, csv_file=csv)
    t0 = time.perf_counter()
    vp.process_video_and_extract_metrics(ref, dist, cfg, csv_file=csv)
    dt = time.perf_counter() - t0
    print("pageable, %2d copier threads: %6.0f frames/s" % (th, n / dt), flush=True)
if os.environ.get("API_PROFILE"):
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(5):
        vp.process_video_and_extract_metrics(dev_r, dev_d, cfg, csv_file=csv)
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(30)
