"""Rate of process_in_batches (complexity_metrics.py:128-148) with this module's kernels on 200 x 1080p host frames handed over
as a LIST of arrays (what the reference's callers hold): every chunk is gathered into one pinned buffer, uploaded once and launched
once, chunk k + 1 gathered while chunk k crosses PCIe."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rtvqa_amd import complexity_metrics as cm, synth
fr = synth.s_natural(200, 1080, 1920, seed=3)
frames = [fr[i] for i in range(len(fr))]
for name, f in (("edge", cm.process_edge_frame), ("dct", cm.process_dct_frame), ("hist", cm.process_histogram_frame)):
    cm.process_in_batches(frames, f, 4, batch_size=50, resize_width=64, resize_height=64)  # warm: scratch, tables, the ring's four slots, the three buffer sets
    t0 = time.perf_counter()
    out = cm.process_in_batches(frames, f, 4, batch_size=50, resize_width=64, resize_height=64)
    dt = time.perf_counter() - t0
    print("process_in_batches(%s): %d frames in %.1f ms = %.0f frames/s" % (name, len(frames), dt * 1e3, len(frames) / dt))
pairs = [(frames[i], frames[i - 1]) for i in range(1, len(frames))]
cm.process_in_batches(pairs, cm.process_frame_complexity, 4, batch_size=50)
t0 = time.perf_counter(); cm.process_in_batches(pairs, cm.process_frame_complexity, 4, batch_size=50); dt = time.perf_counter() - t0
print("process_in_batches(motion): %d pairs in %.1f ms = %.0f pairs/s" % (len(pairs), dt * 1e3, len(pairs) / dt))
