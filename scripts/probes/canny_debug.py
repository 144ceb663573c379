"""Where does the device Canny map differ from the oracle's?  (debugging aid; run on the GPU box)"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import rtvqa_amd  # noqa: E402
from oracle import c_oracle as co  # noqa: E402
from rtvqa_amd import _native as N, synth  # noqa: E402

eng = rtvqa_amd.Engine(0)
for (h, w) in ((64, 64), (70, 100), (64, 300), (130, 600), (64, 1030)):
    fr = synth.s_natural(1, h, w, seed=5)
    rec = eng.complexity(fr, mask=N.M_EDGE)
    g = co.bgr2gray(fr[0])
    cnt, strong, weak, emap = co.canny(g, 100, 200, want_map=True)
    dev = eng.debug_plane(2, 0, h, w) != 0
    bad = np.argwhere(dev != (emap != 0))
    print("%dx%d: oracle (cnt,strong,weak)=(%d,%d,%d) device=(%d,%d,%d) map mismatches=%d dbg=%d" % (
        h, w, cnt, strong, weak, rec[0]["edge_count"], rec[0]["edge_strong"], rec[0]["edge_weak"], len(bad), rec[0]["orb_response"]))
    if len(bad):
        ys, xs = bad[:, 0], bad[:, 1]
        print("   rows: min %d max %d  first 12 (y,x): %s" % (ys.min(), ys.max(), [tuple(map(int, b)) for b in bad[:12]]))
        print("   x%%64 histogram (nonzero): %s" % {int(k): int(v) for k, v in zip(*np.unique(xs % 64, return_counts=True))})
        print("   x//256 histogram: %s" % {int(k): int(v) for k, v in zip(*np.unique(xs // 256, return_counts=True))})
