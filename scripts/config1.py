"""BASELINE config 1 (the reference's own config.json: 64x64 resize, frame_interval=10) end to end on host
frames: 300 x 1080p BGR in host memory -> calculate_average_scene_complexity.  Prints wall time and the tuple."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rtvqa_amd import synth, complexity_metrics as cm
fr = synth.s_natural(300, 1080, 1920, seed=1234)
cm.get_engine()
cm.calculate_average_scene_complexity(fr[:40], 64, 64, frame_interval=10)  # warm-up (allocations, tables)
t0 = time.perf_counter()
out = cm.calculate_average_scene_complexity(fr, 64, 64, frame_interval=10)
dt = time.perf_counter() - t0
print("config1: 300 frames (30 selected) in %.1f ms -> %.0f source frames/s, %.0f selected frames/s" % (dt * 1e3, 300 / dt, 30 / dt))
print(out)
