"""Driver for profiling the full-frame DCT path alone: 5 launches (energy + temporal) on 64 device-resident 1080p frames.
usage: rocprofv3 --kernel-trace --stats -d out -- python3 scripts/dct_full_only.py [h w]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, rtvqa_amd
from rtvqa_amd import _native as N, synth
eng = rtvqa_amd.Engine(0)
h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1080, 1920)
B = 64
fr = synth.s_natural(B + 1, h, w, seed=5)
d = eng.upload(fr)
for _ in range(5):
    rec = eng.complexity(d.slice(1, B + 1), prev0=d.frame(0), mask=N.M_DCT | N.M_TEMPORAL_DCT, dct_mode=N.DCT_FULL)
print(float(rec[3]["dct_energy"]), float(rec[3]["temporal_dct_l1"]))
