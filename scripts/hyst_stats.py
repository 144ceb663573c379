"""Hysteresis statistics on the bench content: relaxation steps per visited tile (diagnostic record field hyst_steps)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, rtvqa_amd
from rtvqa_amd import _native as N, synth
os.environ["VQA_HYST_STATS"] = "1"
eng = rtvqa_amd.Engine(0)
h, w, B = 1080, 1920, 16
for kind in ("natural", "noise"):
    fr = synth.s_natural(B, h, w, seed=1234) if kind == "natural" else synth.s_noise(B, h, w, seed=1234)
    fr = synth.distort(fr) if kind == "natural" else fr
    rec = eng.complexity(fr, mask=N.M_EDGE)
    tiles = ((h + 63) // 64) * ((w + 63) // 64)
    print(kind, "hyst_steps/frame %.0f = %.2f per tile of %d; edges %.0f strong %.0f weak %.0f" % (
        rec["hyst_steps"].mean(), rec["hyst_steps"].mean() / tiles, tiles, rec["edge_count"].mean(), rec["edge_strong"].mean(), rec["edge_weak"].mean()))
