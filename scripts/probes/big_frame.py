"""One 7680x4320 frame pair and one 1x70000 / 70000x1 sliver through every kernel against the oracle (ad-hoc size check)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, rtvqa_amd
from rtvqa_amd import _native as N, synth
from rtvqa_amd.engine import bgr_planes
from oracle import check
eng = rtvqa_amd.Engine(0)
for (h, w) in ((4320, 7680), (3, 20000), (20000, 5), (1100, 4100)):
    fr = synth.s_natural(2, h, w, seed=h + w)
    dist = synth.distort(fr)
    rec = eng.complexity(dist[1:], prev0=dist[0], mask=N.M_ALL, dct_mode=N.DCT_BLOCK8)
    q = eng.quality(fr[1:], dist[1:], bgr_planes(h, w), N.SSIM_GAUSS) if min(h, w) >= 11 else None
    exp = check.expected(fr[1] if q is not None else None, dist[1], dist[0], True, ("gauss",))
    bad = check.compare(exp, rec[0], q[0] if q is not None else None, "gauss")
    print("%dx%d: %s  edges %d" % (w, h, "OK" if not bad else bad, int(rec[0]["edge_count"])), flush=True)
