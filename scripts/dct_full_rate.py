"""Rate of the reference-true full-frame temporal DCT at 1080p (k_gemm_nt_mfma): 8 device-resident frames."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, rtvqa_amd
from rtvqa_amd import _native as N, synth
eng = rtvqa_amd.Engine(0)
h, w, B = 1080, 1920, 8
d = eng.upload(synth.s_natural(B + 1, h, w, seed=5))
p = eng.make_params(dct_mode=N.DCT_FULL)
eng.complexity(d.slice(1, B + 1), prev0=d.frame(0), mask=N.M_TEMPORAL_DCT, params=p)
eng.profile(True)
t0 = time.perf_counter()
for _ in range(3):
    rec = eng.complexity(d.slice(1, B + 1), prev0=d.frame(0), mask=N.M_TEMPORAL_DCT, params=p)
dt = (time.perf_counter() - t0) / 3
flop = 2.0 * (1920 * 1920 * 1080 + 1080 * 1080 * 1920) * B
print("full-frame temporal DCT 1080p: %.2f ms per %d frames = %.0f frames/s, %.1f TFLOP/s fp32 MFMA" % (dt * 1e3, B, B / dt, flop / dt / 1e12))
print(eng.profile_read())
