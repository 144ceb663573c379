"""Run-to-run determinism soak: the same device-resident batch N times; every result record must be bit-identical to the
first run's (all fields except the diagnostic hyst_steps).  Catches races (work lists, hand-offs between hysteresis
rounds), hazards that only bite under some instruction timings, and uninitialised reads.
usage: python scripts/probes/determinism.py [iterations]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, rtvqa_amd
from rtvqa_amd import _native as N, synth
from rtvqa_amd.engine import bgr_planes

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
eng = rtvqa_amd.Engine(0)
for (h, w, B, kind) in ((1080, 1920, 64, "natural"), (1080, 1920, 32, "noise"), (2160, 3840, 8, "natural"), (97, 131, 200, "natural")):
    fr = synth.s_natural(B + 1, h, w, seed=9) if kind == "natural" else synth.s_noise(B + 1, h, w, seed=9)
    dist = synth.distort(fr)
    dr, dd = eng.upload(fr), eng.upload(dist)
    params = eng.make_params(dct_mode=N.DCT_BLOCK8)
    first = None
    bad = 0
    for it in range(iters):
        eng.quality_submit(dr.slice(1, B + 1), dd.slice(1, B + 1), bgr_planes(h, w), N.SSIM_GAUSS)
        eng.complexity_submit(dd.slice(1, B + 1), dd.frame(0), N.M_ALL, params)
        q, c = eng.quality_wait(), eng.complexity_wait()
        cur = tuple(c[f].tobytes() for f in c.dtype.names if f != "hyst_steps") + (q["sse"].tobytes(), q["ssim"].tobytes())
        if first is None:
            first = cur
        elif cur != first:
            bad += 1
            names = [f for f in c.dtype.names if f != "hyst_steps"] + ["sse", "ssim"]
            print("  iteration %d differs in %s" % (it, [n for n, a, b in zip(names, cur, first) if a != b]), flush=True)
    print("%dx%d x %d %s: %d iterations, %d differing" % (w, h, B, kind, iters, bad), flush=True)
    del dr, dd
