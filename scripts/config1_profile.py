import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from rtvqa_amd import synth, complexity_metrics as cm, _native as N
fr = synth.s_natural(300, 1080, 1920, seed=1234)
eng = cm.get_engine()
cm.calculate_average_scene_complexity(fr[:40], 64, 64, frame_interval=10)
eng.profile(True); eng.profile_read(reset=True)
for rep in range(3):
    t0 = time.perf_counter()
    out = cm.calculate_average_scene_complexity(fr, 64, 64, frame_interval=10)
    dt = time.perf_counter() - t0
    prof = eng.profile_read(reset=True)
    print("config1 %.2f ms; kernel ms: %s total %.3f" % (dt * 1e3, {k: round(v[0], 3) for k, v in prof.items()}, sum(v[0] for v in prof.values())))
# pinned variant: the selected frames already in a pinned buffer
sel = eng.alloc_pinned((30, 1080, 1920, 3))
sel[:] = fr[9::10]
prev = sel[0]
for rep in range(3):
    t0 = time.perf_counter()
    rec = eng.complexity(sel[1:], prev0=sel[0], mask=N.M_ALL, resize=(64, 64))
    dt = time.perf_counter() - t0
    print("pinned selected frames, one submit: %.2f ms" % (dt * 1e3))
