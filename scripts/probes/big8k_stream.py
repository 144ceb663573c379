import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from rtvqa_amd import complexity_metrics as cm, stream, video_processing as vp, synth
n, h, w = 6, 4320, 7680
ref = synth.s_natural(n, h, w, seed=9)
dist = synth.distort(ref)
eng = cm.get_engine()
q, cx = stream.Quality(vp.bgr_planes(h, w)), stream.Complexity((64, 64), 2)
dev_r, dev_d = eng.upload(ref), eng.upload(dist)
want_q, want_s = stream.run(dev_d, dev_r, q, cx, batch_size=100, engine=eng)
t0 = time.perf_counter()
got_q, got_s = stream.run(dist, ref, q, cx, batch_size=100)
print("8K pageable fused pass: %.1f ms for %d frames" % ((time.perf_counter() - t0) * 1e3, n))
assert np.array_equal(got_q[0], want_q[0]), 'sse differs'
print('max rel. ssim difference between 6-frame and 2-frame launches: %.3g' % np.max(np.abs(got_q[1] - want_q[1]) / want_q[1]))
assert np.allclose(got_q[1], want_q[1], rtol=1e-6, atol=0)
for k in ("motion", "dct", "hist", "edge", "orb", "color", "temporal"):
    assert len(got_s[k]) == len(want_s[k]) and all(a == b for a, b in zip(got_s[k], want_s[k])), k
print("8K ok", len(got_s["dct"]), got_q[1][0])
