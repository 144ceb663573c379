"""Differential soak of process_in_batches (complexity_metrics.py:128-148): random lists of frames - gray and BGR, a few of another
size, pairs with None frames, chained and unchained pairs - through the batched launches against the per-item callables, value
for value and type for type, in order.
usage: python scripts/fuzz_pib.py [n_cases] [seed0]   (needs a GPU; exits non-zero on the first difference)"""
import functools, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rtvqa_amd import complexity_metrics as cm, synth

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
FUNCS = [cm.process_dct_frame, cm.process_histogram_frame, cm.process_color_histogram_frame, cm.process_edge_frame]


def same(a, b):
    return type(a) is type(b) and (a == b or (a != a and b != b))


for case in range(n_cases):
    r = np.random.default_rng(seed0 + case)
    n, h, w = int(r.integers(1, 30)), int(r.integers(16, 100)), int(r.integers(16, 140))
    batch = int(r.integers(1, 12))
    clip = synth.s_natural(n, h, w, seed=case) if case % 2 else r.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
    frames = [clip[i] for i in range(n)]
    if case % 4 == 1:   # some gray frames (2-D), as the reference's temporal path holds them
        frames = [f[..., 0].copy() if r.random() < 0.3 else f for f in frames]
    if case % 5 == 2:   # one frame of another size: its chunk falls back to the per-item map
        k = int(r.integers(0, n))
        frames[k] = np.ascontiguousarray(frames[k][: h // 2 + 1, : w // 2 + 1])
    rw, rh = int(r.integers(8, 70)), int(r.integers(8, 70))
    f = FUNCS[case % len(FUNCS)]
    got = cm.process_in_batches(frames, f, 3, batch_size=batch, resize_width=rw, resize_height=rh)
    want = [f(x, rw, rh) for x in frames]
    assert len(got) == len(want) and all(same(a, b) for a, b in zip(got, want)), (case, f.__name__, n, h, w, batch)
    got = cm.process_in_batches(frames, cm.process_orb_frame_for_parallel, 3, batch_size=batch)
    assert got == [cm.process_orb_frame_for_parallel(x) for x in frames], (case, "orb")
    # pairs: chained (current, previous), with holes, and shuffled
    col = [x if x.ndim == 3 and x.shape == (h, w, 3) else clip[i] for i, x in enumerate(frames)]
    pairs = [(col[i], col[i - 1]) for i in range(1, n)]
    if case % 3 == 0 and pairs:
        k = int(r.integers(0, len(pairs)))
        pairs[k] = (None, pairs[k][1]) if r.random() < 0.5 else (pairs[k][0], None)
    if case % 3 == 1:
        pairs = [pairs[i] for i in r.permutation(len(pairs))]
    got = cm.process_in_batches(pairs, cm.process_frame_complexity, 3, batch_size=batch)
    want = [cm.process_frame_complexity(p) for p in pairs]
    assert len(got) == len(want) and all(same(a, b) for a, b in zip(got, want)), (case, "motion", n, h, w, batch)
    if case % 50 == 49:
        print("case %d ok" % (case + 1), flush=True)
print("fuzz_pib: %d cases, no difference" % n_cases)
