"""Driver for counter runs: 3 DCT (energy+temporal) launches on 64 device-resident 1080p frames."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, rtvqa_amd
from rtvqa_amd import _native as N, synth
eng = rtvqa_amd.Engine(0)
h, w, B = 1080, 1920, 64
fr = synth.s_natural(B + 1, h, w, seed=5)
d = eng.upload(fr)
for _ in range(3):
    eng.complexity(d.slice(1, B + 1), prev0=d.frame(0), mask=N.M_DCT | N.M_TEMPORAL_DCT, dct_mode=N.DCT_BLOCK8)
