"""The reference-shaped entry point under a profiler: process_video_and_extract_metrics on a 1080p clip resident in HBM, from
pinned host memory and from pageable host memory (the three legs of bench.py's api_end_to_end), a few calls each - so that
`rocprofv3 --kernel-trace --stats -- python3 scripts/api_trace.py` shows WHICH kernels the drop-in surface launches and how
long they take there (profiles/round5_api_kernel_stats.csv): the same k_ssim_gauss_p2 / k_block_sad / k_canny_* /
k_bgr2gray_hist / k_dct8_march as the C-ABI bench, in launches of <= 100 frames."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rtvqa_amd import complexity_metrics as cm, synth, video_processing as vp

n, h, w = int(sys.argv[1]) if len(sys.argv) > 1 else 129, 1080, 1920
eng = cm.get_engine()
pin_r, pin_d = eng.alloc_pinned((n, h, w, 3)), eng.alloc_pinned((n, h, w, 3))
for a in range(0, n, 32):
    r = synth.s_natural(min(32, n - a), h, w, seed=1234, t0=a)
    pin_r[a:a + len(r)] = r
    pin_d[a:a + len(r)] = synth.distort(r, t0=a)
ref, dist = np.array(pin_r), np.array(pin_d)
dev_r, dev_d = eng.upload(ref), eng.upload(dist)
cfg = {"crf": 23, "resize_width": w, "resize_height": h, "frame_interval": 1, "batch_size": 100}
csv = os.path.join(tempfile.gettempdir(), "api_trace.csv")
for name, (r, d) in (("resident", (dev_r, dev_d)), ("host_pinned", (pin_r, pin_d)), ("host_pageable", (ref, dist))):
    t0 = time.perf_counter()
    for _ in range(3):
        m = vp.process_video_and_extract_metrics(r, d, cfg, csv_file=csv)
    print("%-14s 3 calls x %d frames in %.1f ms (first call included)  PSNR %.2f SSIM %.6f" % (name, n, (time.perf_counter() - t0) * 1e3, m["PSNR"], m["SSIM"]), flush=True)
cm.release_buffers()
