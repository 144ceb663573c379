"""Differential soak of the one-pass host side (stream.run): random clip length / geometry / frame_interval / chunk size /
residence (pageable, pinned, region of interest, memmap, device) and random halves (quality only, complexity only, fused, and the
split pass: a planar yuv420p quality pair next to the BGR stream, each in its own place) - every combination must return the bits
of the one-chunk, one-engine pass over a contiguous copy of the clip.
usage: python scripts/fuzz_stream.py [n_cases] [seed0]   (needs a GPU; exits non-zero on the first difference)"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rtvqa_amd import complexity_metrics as cm, stream, synth
from rtvqa_amd.engine import bgr_planes

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
eng = cm.get_engine()
tmp = tempfile.mkdtemp(prefix="vqa_fuzz_")
KINDS = ("motion", "dct", "hist", "edge", "orb", "color", "temporal")
for case in range(n_cases):
    r = np.random.default_rng(seed0 + case)
    n, h, w = int(r.integers(1, 40)), int(r.integers(11, 120)), int(r.integers(11, 160))
    iv, batch = int(r.integers(1, 7)), int(r.integers(1, 12))
    if case % 25 == 24:  # a long clip of small frames: dozens of chunks, both lanes and every ring slot many times over
        n, h, w, batch = int(r.integers(200, 900)), int(r.integers(11, 64)), int(r.integers(11, 64)), int(r.integers(3, 40))
    rw, rh = (int(r.integers(8, 80)), int(r.integers(8, 80))) if case % 2 else (w, h)
    ref = synth.s_natural(n, h, w, seed=case) if case % 3 else r.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
    dist = synth.distort(ref)
    res = int(r.integers(0, 5))
    keep = []
    if res == 0:
        a, b = ref, dist
    elif res == 1:
        a, b = eng.alloc_pinned(ref.shape), eng.alloc_pinned(dist.shape)
        a[...], b[...] = ref, dist
        keep = [a, b]
    elif res == 2:  # a region of interest inside larger frames (padded rows)
        big_r, big_d = np.zeros((n, h + 5, w + 9, 3), np.uint8), np.zeros((n, h + 5, w + 9, 3), np.uint8)
        big_r[:, 2:2 + h, 4:4 + w], big_d[:, 2:2 + h, 4:4 + w] = ref, dist
        a, b = big_r[:, 2:2 + h, 4:4 + w], big_d[:, 2:2 + h, 4:4 + w]
    elif res == 3:
        np.save(os.path.join(tmp, "r.npy"), ref); np.save(os.path.join(tmp, "d.npy"), dist)
        a, b = np.load(os.path.join(tmp, "r.npy"), mmap_mode="r"), np.load(os.path.join(tmp, "d.npy"), mmap_mode="r")
    else:
        a, b = eng.upload(ref), eng.upload(dist)
    half = int(r.integers(0, 3))
    q = stream.Quality(bgr_planes(h, w)) if half != 1 else None
    cx = stream.Complexity((rw, rh), iv) if half != 0 else None
    want_q, want_s = stream.run(dist, ref if q else None, q, cx, batch_size=10 ** 6, engine=eng)
    got_q, got_s = stream.run(b, a if q else None, q, cx, batch_size=batch)
    ctx = (case, n, h, w, iv, batch, rw, rh, res, half)
    if q:
        assert np.array_equal(got_q[0], want_q[0]) and np.array_equal(got_q[1], want_q[1]), ("quality", ctx)
    if cx:
        for k in KINDS:
            assert len(got_s[k]) == len(want_s[k]) and all((x == y) or (x != x and y != y) for x, y in zip(got_s[k], want_s[k])), (k, ctx)
    if case % 4 == 0:
        # the quality half on the other layouts (gray planes, planar yuv420p with odd sizes) and FFmpeg's integer SSIM, from
        # pageable and pinned memory: chunked against one chunk
        from rtvqa_amd import frames as fr_, video_processing as vp
        lay = ("gray", "yuv420p")[case // 4 % 2]
        mode = ("gauss", "ffmpeg")[case // 8 % 2]
        hh, ww = max(h, 16), max(w, 16)
        if lay == "gray":
            qa, qb = np.ascontiguousarray(ref[..., 0]), np.ascontiguousarray(dist[..., 0])
            if (hh, ww) != (h, w):
                qa, qb = np.pad(qa, ((0, 0), (0, hh - h), (0, ww - w))), np.pad(qb, ((0, 0), (0, hh - h), (0, ww - w)))
        else:
            pr = np.pad(ref, ((0, 0), (0, hh - h), (0, ww - w), (0, 0))) if (hh, ww) != (h, w) else ref
            pd = np.pad(dist, ((0, 0), (0, hh - h), (0, ww - w), (0, 0))) if (hh, ww) != (h, w) else dist
            qa, qb = fr_.bgr_to_yuv420p(pr), fr_.bgr_to_yuv420p(pd)
            if min((hh + 1) // 2, (ww + 1) // 2) < 11:
                mode = "ffmpeg" if min((hh + 1) // 2, (ww + 1) // 2) >= 8 else None
        if mode:
            one = vp.frame_quality(qa, qb, lay, mode, hh, ww, engine=eng, batch_size=10 ** 6)
            pa, pb = eng.alloc_pinned(qa.shape), eng.alloc_pinned(qb.shape)
            pa[...], pb[...] = qa, qb
            for x, y in ((qa, qb), (pa, pb)):
                got = vp.frame_quality(x, y, lay, mode, hh, ww, batch_size=batch)
                assert np.array_equal(got[0], one[0]) and np.array_equal(got[1], one[1]), ("layout", lay, mode, (case, n, hh, ww, batch))
            eng.free_pinned(pa); eng.free_pinned(pb)
    if case % 3 == 0 and min(h, w) >= 16:
        # the SPLIT pass (the reference's own quantities: a planar yuv420p quality pair next to the encoded BGR stream, round 6):
        # the three streams in three random places, chunked, against the two halves run alone in one chunk each
        from rtvqa_amd import _native as N_, frames as fr_
        from rtvqa_amd.engine import DeviceFrames, yuv420p_planes
        ya, yb = fr_.bgr_to_yuv420p(ref), fr_.bgr_to_yuv420p(dist)
        ypl = yuv420p_planes(h, w)
        smode = N_.SSIM_FFMPEG if (case // 3) % 2 or min((h + 1) // 2, (w + 1) // 2) < 11 else N_.SSIM_GAUSS
        cxs = stream.Complexity((rw, rh), iv)
        one_q, _ = stream.run(yb, ya, stream.Quality(ypl, smode), batch_size=10 ** 6, engine=eng)
        _, one_s = stream.run(dist, complexity=cxs, batch_size=10 ** 6, engine=eng)
        held, pins = [], []

        def place(arr, planar, where):
            if where == 0:
                return arr
            if where == 1:
                pp = eng.alloc_pinned(arr.shape)
                pp[...] = arr
                pins.append(pp)
                return pp
            d = eng.upload(arr.reshape(n, 1, -1) if planar else arr)
            held.append(d)
            return DeviceFrames(d.ptr, n, h, w, frame_stride=arr.shape[1], row_stride=w, owner=d, channels=1) if planar else d
        wq = int(r.integers(0, 3))   # (the quality pair shares a residence class: device with device)
        xs = place(dist, False, int(r.integers(0, 3))), place(ya, True, wq), place(yb, True, wq if wq == 2 else int(r.integers(0, 2)))
        got_q, got_s = stream.run(xs[0], xs[1], stream.Quality(ypl, smode), cxs, batch_size=batch, qdist=xs[2])
        assert np.array_equal(got_q[0], one_q[0]) and np.array_equal(got_q[1], one_q[1]), ("split quality", ctx)
        for k in KINDS:
            assert len(got_s[k]) == len(one_s[k]) and all((x == y) or (x != x and y != y) for x, y in zip(got_s[k], one_s[k])), ("split", k, ctx)
        for pp in pins:
            eng.free_pinned(pp)
        for d in held:
            d._owner.free()
    for p in keep:
        eng.free_pinned(p)
    if res == 4:
        a._owner.free(); b._owner.free()
    if case % 100 == 99:
        print("case %d ok" % (case + 1), flush=True)
        cm.release_buffers()
print("fuzz_stream: %d cases, no difference" % n_cases)
