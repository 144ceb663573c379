"""Driver for counter runs: 3 SSIM launches on 64 device-resident 1080p BGR pairs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, rtvqa_amd
from rtvqa_amd import _native as N, synth
from rtvqa_amd.engine import bgr_planes
eng = rtvqa_amd.Engine(0)
h, w, B = 1080, 1920, 64
ref = synth.s_natural(B, h, w, seed=5); dist = synth.distort(ref)
b_r = eng.upload(ref); b_d = eng.upload(dist)
for _ in range(3):
    eng.quality(b_r, b_d, bgr_planes(h, w), N.SSIM_GAUSS)
