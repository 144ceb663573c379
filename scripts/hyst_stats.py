"""Hysteresis statistics on the bench content: relaxation iterations per tile (diagnostic record field hyst_steps,
filled only with VQA_OPT_HYST_STATS), at 1080p (workload c3) and 2160p (c4): the same number of tiles per launch
(256 x 510 = 64 x 2040) but different work per tile."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, rtvqa_amd
from rtvqa_amd import _native as N, synth
eng = rtvqa_amd.Engine(0)
eng.set_option(N.OPT_HYST_STATS, 1)
for (h, w, B) in ((1080, 1920, 16), (2160, 3840, 4)):
    for kind in ("natural", "noise"):
        fr = synth.s_natural(B, h, w, seed=1234) if kind == "natural" else synth.s_noise(B, h, w, seed=1234)
        fr = synth.distort(fr) if kind == "natural" else fr
        rec = eng.complexity(fr, mask=N.M_EDGE)
        tiles = ((h + 63) // 64) * ((w + 63) // 64)
        print("%dx%d %-7s iterations/frame %.0f = %.2f per tile of %d; edges %.0f strong %.0f weak %.0f (%.2f %% of pixels are candidates)" % (
            w, h, kind, rec["hyst_steps"].mean(), rec["hyst_steps"].mean() / tiles, tiles, rec["edge_count"].mean(),
            rec["edge_strong"].mean(), rec["edge_weak"].mean(), 100.0 * (rec["edge_strong"].mean() + rec["edge_weak"].mean()) / (h * w)), flush=True)
