"""Driver for profiling the Farneback pyramid alone: 3 launches of 32 device-resident 1080p pairs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, rtvqa_amd
from rtvqa_amd import _native as N, synth
eng = rtvqa_amd.Engine(0)
h, w, B = 1080, 1920, 32
fr = synth.s_natural(B + 1, h, w, seed=5)
d = eng.upload(fr)
for rep in range(3):
    t0 = time.perf_counter()
    rec = eng.complexity(d.slice(1, B + 1), prev0=d.frame(0), mask=N.M_MOTION, motion_mode=N.MOTION_FARNEBACK)
    dt = time.perf_counter() - t0
    print("farneback %d pairs 1080p: %.2f ms (%.3f ms/pair), mean |flow| of pair 0 = %.4f" % (B, dt * 1e3, dt * 1e3 / B, rec[0]["flow_mag_mean"]), flush=True)
