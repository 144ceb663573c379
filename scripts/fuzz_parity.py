"""One-off differential soak: random geometries / thresholds / ranges through every kernel against the oracle.
usage: python scripts/fuzz_parity.py [n_cases] [seed0]   (needs a GPU; exits non-zero on the first mismatch)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rtvqa_amd
from rtvqa_amd import _native as N, synth
from rtvqa_amd.engine import bgr_planes
from oracle import c_oracle as co

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
eng = rtvqa_amd.Engine(0)
RT = 1e-4
for case in range(n_cases):
    r = np.random.default_rng(seed0 + case)
    h, w = int(r.integers(1, 200)), int(r.integers(1, 320))
    kind = int(r.integers(0, 3))
    n = int(r.integers(1, 4))
    if kind == 0:
        fr = r.integers(0, 256, (n + 1, h, w, 3), dtype=np.uint8)
    elif kind == 1:
        fr = synth.s_natural(n + 1, h, w, seed=case)
    else:
        fr = np.repeat(r.integers(0, 256, (n + 1, (h + 7) // 8, (w + 7) // 8, 3), dtype=np.uint8), 8, axis=1).repeat(8, axis=2)[:, :h, :w]
    lo, hi = sorted(int(x) for x in r.integers(1, 300, 2))
    rng_ = int(r.integers(0, 8))
    use_fb = case % 5 == 0
    rw, rh = (int(r.integers(1, 100)), int(r.integers(1, 100))) if case % 3 == 0 else (w, h)
    rec = eng.complexity(fr[1:], prev0=fr[0], mask=N.M_ALL, canny=(lo, hi), sad_range=rng_, resize=(rw, rh),
                         dct_mode=N.DCT_BLOCK8, motion_mode=N.MOTION_FARNEBACK if use_fb else N.MOTION_SAD)
    for i in range(n):
        ctx = (case, h, w, rw, rh, lo, hi, rng_, kind, i)
        rs = co.resize_linear(fr[i + 1], rw, rh)
        gb = co.bgr2gray(rs)
        ga, gpa = co.resize_linear(co.bgr2gray(fr[i + 1]), rw, rh), co.resize_linear(co.bgr2gray(fr[i]), rw, rh)
        assert (rec[i]["hist_gray"] == co.hist_u8(gb)).all(), ("hist", ctx)
        for c in range(3):
            assert (rec[i]["hist_bgr"][c] == co.hist_u8(rs, offset=c, step=3)).all(), ("chist", ctx)
        cnt, st, wk = co.canny(gb, lo, hi)
        assert (int(rec[i]["edge_count"]), int(rec[i]["edge_strong"]), int(rec[i]["edge_weak"])) == (cnt, st, wk), ("canny", ctx)
        e, l1, _ = co.dct8x8(gpa, ga)
        assert e == 0 or abs(rec[i]["dct_energy"] - e) <= RT * e, ("dct", ctx)
        assert l1 == 0 or abs(rec[i]["temporal_dct_l1"] - l1) <= RT * l1, ("tdct", ctx)
        g, gp = co.bgr2gray(fr[i + 1]), co.bgr2gray(fr[i])
        if use_fb:
            want = co.farneback(gp, g)
            assert abs(rec[i]["flow_mag_mean"] - want) <= RT * want + 1e-7, ("farneback", ctx, rec[i]["flow_mag_mean"], want)
        else:
            nb, sad, hist = co.block_sad(gp, g, rng_)
            assert (int(rec[i]["sad_sum"]), int(rec[i]["sad_blocks"])) == (sad, nb) and (rec[i]["mv_d2_hist"] == hist).all(), ("sad", ctx)
        assert int(rec[i]["orb_keypoints"]) == co.orb64_count(co.bgr2gray(co.resize_linear(fr[i + 1], 64, 64)))[0], ("orb", ctx)
    if case % 7 == 0:  # the full-frame cv2.dct parity mode: VALU tiles below 128x128, MFMA tiles above
        fw, fh = int(r.integers(1, 260)), int(r.integers(1, 200))
        rec = eng.complexity(fr[1:], prev0=fr[0], mask=N.M_DCT | N.M_TEMPORAL_DCT, resize=(fw, fh), dct_mode=N.DCT_FULL)
        for i in range(n):
            a, p_ = co.resize_linear(co.bgr2gray(fr[i + 1]), fw, fh), co.resize_linear(co.bgr2gray(fr[i]), fw, fh)
            e, l1 = co.dct_energy_full(a), co.temporal_dct_full(p_, a)
            assert e == 0 or abs(rec[i]["dct_energy"] - e) <= RT * e, ("dct_full", case, h, w, fw, fh, rec[i]["dct_energy"], e)
            assert l1 == 0 or abs(rec[i]["temporal_dct_l1"] - l1) <= RT * l1, ("tdct_full", case, h, w, fw, fh, rec[i]["temporal_dct_l1"], l1)
    if case % 11 == 0 and h > 8 and w > 8:  # region of interest in place, host and device resident
        y0, x0 = int(r.integers(0, h // 2)), int(r.integers(0, w // 2))
        y1, x1 = int(r.integers(y0 + 1, h + 1)), int(r.integers(x0 + 1, w + 1))
        sub = fr[:, y0:y1, x0:x1]
        want = eng.complexity(np.ascontiguousarray(sub[1:]), prev0=np.ascontiguousarray(sub[0]), mask=N.M_ALL, dct_mode=N.DCT_BLOCK8)
        dev = eng.upload(fr).roi(y0, y1, x0, x1)
        for got in (eng.complexity(sub[1:], prev0=sub[0], mask=N.M_ALL, dct_mode=N.DCT_BLOCK8),
                    eng.complexity(dev.slice(1, n + 1), prev0=dev.frame(0), mask=N.M_ALL, dct_mode=N.DCT_BLOCK8)):
            for f in want.dtype.names:
                assert f == "hyst_steps" or (got[f] == want[f]).all(), ("roi", case, h, w, y0, y1, x0, x1, f)
    if h >= 11 and w >= 11:
        for mode, fn in ((N.SSIM_GAUSS, co.ssim_gauss), (N.SSIM_FFMPEG, co.ssim_ffmpeg)):
            q = eng.quality(fr[:-1], fr[1:], bgr_planes(h, w), mode)
            for i in (0, n - 1):
                for c in range(3):
                    a, b = fr[i][..., c], fr[i + 1][..., c]
                    assert int(q[i, c]["sse"]) == co.sse_plane(a, b), ("sse", case, i, c)
                    want = fn(a, b)
                    # uncorrelated noise frames: the plane mean cancels to ~1e-4..1e-6 (unit scale), hence the 1e-6 absolute floor
                    assert abs(q[i, c]["ssim"] - want) <= RT * abs(want) + 1e-6, ("ssim", mode, case, h, w, i, c, q[i, c]["ssim"], want)
    if case % 25 == 24:
        print("  %d cases ok" % (case + 1), flush=True)
print("fuzz ok: %d cases" % n_cases)
