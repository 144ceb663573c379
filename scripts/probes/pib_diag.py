import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from rtvqa_amd import complexity_metrics as cm, synth, stream, _native as N
fr = synth.s_natural(200, 1080, 1920, seed=3)
frames = [fr[i] for i in range(len(fr))]
eng = cm.get_engine()
pin = eng.alloc_pinned(fr[:50].shape); pin[...] = fr[:50]
for mask, name in ((N.M_EDGE, "edge"), (N.M_GRAY_HIST, "hist"), (N.M_MOTION, "motion")):
    p = eng.make_params(resize=(64, 64))
    eng.complexity(pin, mask=mask, params=p)
    t0 = time.perf_counter()
    for _ in range(4):
        eng.complexity(pin, mask=mask, params=p)
    print("C ABI host-pointer submit, 50 frames, %s: %.1f ms per chunk" % (name, (time.perf_counter() - t0) / 4 * 1e3), flush=True)
    dev = eng.upload(fr[:50])
    eng.complexity(dev, mask=mask, params=p)
    eng.profile(True); eng.profile_read(reset=True)
    t0 = time.perf_counter()
    for _ in range(4):
        eng.complexity(dev, mask=mask, params=p)
    print("   resident: %.2f ms per chunk" % ((time.perf_counter() - t0) / 4 * 1e3), eng.profile_read(reset=True), flush=True)
    eng.profile(False)
for lanes in (1, 2):
    stream.MAX_LANES = lanes
    for name, f in (("edge", cm.process_edge_frame), ("hist", cm.process_histogram_frame)):
        cm.process_in_batches(frames, f, 4, batch_size=50, resize_width=64, resize_height=64)
        t0 = time.perf_counter()
        cm.process_in_batches(frames, f, 4, batch_size=50, resize_width=64, resize_height=64)
        print("lanes %d process_in_batches(%s): %.1f ms" % (lanes, name, (time.perf_counter() - t0) * 1e3), flush=True)
