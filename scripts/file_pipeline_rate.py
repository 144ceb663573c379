"""End-to-end rate of the reference-shaped calls on a HOST-resident 1080p clip (pageable memory, as a caller
that decoded a file would hold it): run_ffmpeg_metrics-style frame_quality and complexity_series, one engine
vs the two-engine ping-pong the library uses for clips longer than a batch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rtvqa_amd import complexity_metrics as cm, synth, video_processing as vp

h, w, n = 1080, 1920, 192
ref = synth.s_natural(n, h, w, seed=5)
dist = synth.distort(ref)
eng = cm.get_engine()
for name, fn in (
        ("frame_quality one engine", lambda: vp.frame_quality(ref, dist, "bgr24", "gauss", batch_size=32, engine=eng)),
        ("frame_quality two engines", lambda: vp.frame_quality(ref, dist, "bgr24", "gauss", batch_size=32)),
        ("complexity_series one engine", lambda: cm.complexity_series(dist, 1920, 1080, 1, batch_size=32, engine=eng)),
        ("complexity_series two engines", lambda: cm.complexity_series(dist, 1920, 1080, 1, batch_size=32))):
    fn()
    t0 = time.perf_counter()
    fn()
    dt = time.perf_counter() - t0
    print("%-32s %6.0f frames/s (%d frames in %.1f ms)" % (name, n / dt, n, dt * 1e3), flush=True)
