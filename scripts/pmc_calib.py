"""Known-byte-count launches for calibrating FETCH_SIZE on this access pattern (run under rocprofv3 --pmc).
1: SSIM on 64 contiguous GRAY plane pairs (1 B/lane loads, step 1): algorithmic read = 2*P*64 bytes
2: SSIM on 64 packed-BGR pairs, 3 channels in one launch (1 B/lane, step 3): algorithmic read = 2*3P*64 bytes"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, rtvqa_amd
from rtvqa_amd import _native as N, synth
from rtvqa_amd.engine import gray_planes, bgr_planes
eng = rtvqa_amd.Engine(0)
h, w, B = 1080, 1920, 64
ref = synth.s_natural(B, h, w, seed=5); dist = synth.distort(ref)
g_r = eng.upload(np.ascontiguousarray(ref[..., 1])); g_d = eng.upload(np.ascontiguousarray(dist[..., 1]))
b_r = eng.upload(ref); b_d = eng.upload(dist)
eng.quality(g_r, g_d, gray_planes(h, w), N.SSIM_GAUSS)
eng.quality(b_r, b_d, bgr_planes(h, w), N.SSIM_GAUSS)
eng.quality(g_r, g_d, gray_planes(h, w), N.SSIM_FFMPEG)
print("P*B =", h * w * B)
