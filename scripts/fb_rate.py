"""The Farneback pyramid alone, timed with HIP events through the engine's profiler: B device-resident 1080p pairs per launch,
a few launches, per-kernel-group ms (VQA_LIB_PATH selects a probe build: scripts/build_probes.sh FB_PROBE ...).
usage: fb_rate.py [pairs=64] [launches=6]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, rtvqa_amd
from rtvqa_amd import _native as N, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
eng = rtvqa_amd.Engine(0)
eng.set_overlap(False)
h, w = 1080, 1920
fr = synth.s_natural(B + 1, h, w, seed=5)
d = eng.upload(fr)
p = eng.make_params(motion_mode=N.MOTION_FARNEBACK)
for _ in range(2):
    rec = eng.complexity(d.slice(1, B + 1), prev0=d.frame(0), mask=N.M_MOTION, params=p)
eng.profile(True); eng.profile_read(reset=True)
t0 = time.perf_counter()
for _ in range(reps):
    rec = eng.complexity(d.slice(1, B + 1), prev0=d.frame(0), mask=N.M_MOTION, params=p)
dt = (time.perf_counter() - t0) / reps
prof = eng.profile_read(reset=True)
print("%s: %d pairs 1080p %.3f ms per launch  %s  mean|flow|[0]=%.9g" % (os.path.basename(os.environ.get("VQA_LIB_PATH", "shipped")), B, dt * 1e3,
      {k: round(ms / reps, 3) for k, (ms, cnt) in prof.items()}, rec[0]["flow_mag_mean"]), flush=True)
