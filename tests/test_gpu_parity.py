"""GPU parity: every HIP kernel, called through the C ABI, against the CPU oracle.

Bit-exact for histogram bins, gray/resized planes, edge counts/maps, SAD sums and SSE;
1e-4 relative (the tolerance BASELINE.json's north_star states) for DCT / SSIM floats.
"""
import os

import numpy as np
import pytest

from oracle import c_oracle as co
from oracle import np_oracle as no
from oracle import pipeline as pl

pytestmark = pytest.mark.gpu

RTOL = 1e-4  # north_star: "within 1e-4 relative for DCT/SSIM/PSNR floats"


def _rng(seed):
    return np.random.default_rng(seed)


REPO_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _frames(kind, n, h, w, seed=0):
    from rtvqa_amd import synth
    if kind == "noise":
        return synth.s_noise(n, h, w, seed=seed)
    if kind == "natural":
        return synth.s_natural(n, h, w, seed=seed)
    raise KeyError(kind)


def _stable(rec):
    """A result batch as comparable values: every field except the diagnostic, scheduling-dependent
    hyst_steps (field by field: the record has alignment padding that NumPy copies do not preserve)."""
    return tuple(rec[f].tobytes() for f in rec.dtype.names if f != "hyst_steps")


def _rel(a, b):
    return abs(a - b) / max(abs(b), 1e-30)


@pytest.fixture(autouse=True)
def _no_hysteresis_overflow(engine, monkeypatch):
    """Every complexity batch of every test: the Canny fixpoint was reached (hyst_overflow is the record's flag for
    "the bounded tail stopped early, edge_count is an under-count")."""
    wait = engine.complexity_wait

    def checked():
        rec = wait()
        assert not rec["hyst_overflow"].any(), "Canny hysteresis stopped at its round bound (2 = the rescue pass had to finish it)"
        return rec
    monkeypatch.setattr(engine, "complexity_wait", checked)


# ---------------------------------------------------------------------------
@pytest.mark.parametrize("h,w", [(48, 64), (37, 53), (130, 208), (16, 16)])
def test_gray_plane_and_histograms_native(engine, h, w):
    from rtvqa_amd import _native as N
    fr = _frames("noise", 3, h, w, seed=h * w)
    rec = engine.complexity(fr, mask=N.M_GRAY_HIST | N.M_COLOR_HIST | N.M_DCT, dct_mode=N.DCT_BLOCK8)
    for i in range(3):
        g = co.bgr2gray(fr[i])
        assert (engine.debug_plane(3, i, h, w) == g).all()
        assert (rec[i]["hist_gray"] == co.hist_u8(g)).all()
        for c in range(3):
            assert (rec[i]["hist_bgr"][c] == co.hist_u8(fr[i], offset=c, step=3)).all()
        assert int(rec[i]["sum_gray2"]) == int((g.astype(np.int64) ** 2).sum())
        assert int(rec[i]["hist_gray"].sum()) == h * w


def test_histograms_degenerate_frames(engine):
    from rtvqa_amd import _native as N
    from rtvqa_amd import synth
    deg = synth.s_degenerate(64, 96)
    names = sorted(deg)
    fr = np.stack([deg[k] for k in names])
    rec = engine.complexity(fr, mask=N.M_GRAY_HIST | N.M_COLOR_HIST)
    for i, k in enumerate(names):
        g = co.bgr2gray(fr[i])
        assert (rec[i]["hist_gray"] == co.hist_u8(g)).all(), k
        assert (rec[i]["hist_bgr"][1] == co.hist_u8(fr[i], offset=1, step=3)).all(), k


@pytest.mark.parametrize("h,w,rw,rh", [(270, 480, 64, 64), (97, 131, 33, 17), (96, 128, 64, 48), (60, 80, 160, 120),
                                       (1080, 1920, 64, 64)])
def test_resize_planes_and_histograms(engine, h, w, rw, rh):
    from rtvqa_amd import _native as N
    fr = _frames("noise" if h < 1000 else "natural", 2, h, w, seed=rw)
    rec = engine.complexity(fr, mask=N.M_GRAY_HIST | N.M_COLOR_HIST | N.M_DCT | N.M_EDGE, resize=(rw, rh))
    for i in range(2):
        a = co.resize_linear(co.bgr2gray(fr[i]), rw, rh)        # :358-359 gray -> resize
        rb = co.resize_linear(fr[i], rw, rh)                     # :404 resize -> gray
        b = co.bgr2gray(rb)
        assert (engine.debug_plane(0, i, rh, rw) == a).all()
        assert (engine.debug_plane(1, i, rh, rw) == b).all()
        assert (rec[i]["hist_gray"] == co.hist_u8(b)).all()
        for c in range(3):
            assert (rec[i]["hist_bgr"][c] == co.hist_u8(rb, offset=c, step=3)).all()
        assert int(rec[i]["sum_gray2"]) == int((a.astype(np.int64) ** 2).sum())
        assert int(rec[i]["edge_count"]) == co.canny(b, 100, 200)[0]


# ---------------------------------------------------------------------------
@pytest.mark.parametrize("h,w", [(64, 64), (72, 88), (37, 51), (8, 8), (130, 520)])
def test_dct8_energy_and_temporal(engine, h, w):
    from rtvqa_amd import _native as N
    fr = _frames("natural", 4, h, w, seed=7)
    rec = engine.complexity(fr[1:], prev0=fr[0], mask=N.M_DCT | N.M_TEMPORAL_DCT, dct_mode=N.DCT_BLOCK8)
    for i in range(3):
        g, gp = co.bgr2gray(fr[i + 1]), co.bgr2gray(fr[i])
        e, l1, _ = co.dct8x8(gp, g)
        assert _rel(rec[i]["dct_energy"], e) < RTOL, (rec[i]["dct_energy"], e)
        assert _rel(rec[i]["dct_energy"], float(rec[i]["sum_gray2"])) < RTOL  # Parseval known answer
        assert _rel(rec[i]["temporal_dct_l1"], l1) < RTOL, (rec[i]["temporal_dct_l1"], l1)
        assert rec[i]["has_prev"] == 1
    # without a previous frame the first temporal sample is 0 and flagged
    rec = engine.complexity(fr, mask=N.M_TEMPORAL_DCT, dct_mode=N.DCT_BLOCK8)
    assert rec[0]["temporal_dct_l1"] == 0.0 and rec[0]["has_prev"] == 0 and rec[1]["temporal_dct_l1"] > 0


def test_dct8_constant_frames_known_answer(engine):
    from rtvqa_amd import _native as N
    a = np.full((1, 64, 96, 3), 100, np.uint8)
    b = np.full((64, 96, 3), 97, np.uint8)
    rec = engine.complexity(a, prev0=b, mask=N.M_DCT | N.M_TEMPORAL_DCT, dct_mode=N.DCT_BLOCK8)
    # dct(prev) - dct(curr) in fp32, as the reference computes it (:574-578): the two DC terms (776, 800) round
    # separately, so the known answer holds to a few fp32 ulps of the DC term, not of the difference
    assert _rel(rec[0]["temporal_dct_l1"], 3 * 64 * 96 / 8) < 2e-5
    assert _rel(rec[0]["dct_energy"], 100.0 ** 2 * 64 * 96) < 1e-6
    rec = engine.complexity(a, prev0=a[0], mask=N.M_TEMPORAL_DCT, dct_mode=N.DCT_BLOCK8)
    assert rec[0]["temporal_dct_l1"] == 0.0


@pytest.mark.parametrize("h,w", [(1080, 1920), (1078, 1918), (720, 1280), (1440, 2560), (2160, 3840)])
def test_dct_full_frame_1080p_native_vs_scipy(engine, h, w):
    """N1: the reference's full-frame temporal DCT at NATIVE resolution (complexity_metrics.py:363-364, :574-579).
    1080x1920 and 720x1280 factor into 2, 3, 5 and take the FFT-based row / column passes (k_dct_fft.hip: rows in place,
    eight-column slabs; 1440x2560 = the larger register class of both, 2160x3840 = 512-thread rows and the
    pair-per-workgroup column kernel); 1078x1918
    (= 2 7^2 11 x 2 7 137) does not and takes the fp32 MFMA products (k_dct_full.hip: 128x128 tiles with a ragged edge, the
    prefetch tail and the |.| reduction).  Checker: scipy.fft.dctn(norm="ortho") in float64 (the C oracle's O(N^3) loop
    is too slow at this size)."""
    import scipy.fft
    from rtvqa_amd import _native as N
    fr = _frames("natural", 3, h, w, seed=23)
    rec = engine.complexity(fr[1:], prev0=fr[0], mask=N.M_DCT | N.M_TEMPORAL_DCT, dct_mode=N.DCT_FULL)
    coef = [scipy.fft.dctn(co.bgr2gray(f).astype(np.float64), norm="ortho") for f in fr]
    for i in range(2):
        e = float((coef[i + 1] ** 2).sum())
        l1 = float(np.abs(coef[i] - coef[i + 1]).sum())
        assert _rel(rec[i]["dct_energy"], e) < RTOL, (rec[i]["dct_energy"], e)
        assert _rel(rec[i]["dct_energy"], float(rec[i]["sum_gray2"])) < RTOL  # Parseval
        assert _rel(rec[i]["temporal_dct_l1"], l1) < RTOL, (rec[i]["temporal_dct_l1"], l1)
    # a constant brightness step: full-frame L1 = |delta| * sqrt(W H) exactly (only the DC term moves)
    a = np.full((1, h, w, 3), 120, np.uint8)
    rec = engine.complexity(a, prev0=np.full((h, w, 3), 117, np.uint8), mask=N.M_TEMPORAL_DCT, dct_mode=N.DCT_FULL)
    assert _rel(rec[0]["temporal_dct_l1"], 3 * np.sqrt(h * w)) < RTOL


@pytest.mark.parametrize("h,w", [(270, 480), (1080, 1920), (134, 262)])
def test_dct_full_frame_static_and_nearly_static_scenes(engine, h, w):
    """The temporal full-frame metric on a scene that does not move: identical frames must give EXACTLY 0 (the reference
    subtracts two identical coefficient arrays), and a frame that differs from its predecessor in a handful of pixels
    must keep the 1e-4 bar although its L1 is six orders of magnitude below the plane's own coefficients.  The first
    FFT version packed the plane and the difference into one complex transform and read 0.5 instead of 0 here: rounding
    noise of the large plane leaks into the small difference.  (134 x 262 takes the dense products.)"""
    import scipy.fft
    from rtvqa_amd import _native as N
    fr = _frames("natural", 1, h, w, seed=57)
    same = np.repeat(fr, 3, axis=0)
    rec = engine.complexity(same[1:], prev0=same[0], mask=N.M_DCT | N.M_TEMPORAL_DCT, dct_mode=N.DCT_FULL)
    assert float(rec[0]["temporal_dct_l1"]) == 0.0 and float(rec[1]["temporal_dct_l1"]) == 0.0
    assert _rel(rec[0]["dct_energy"], float(rec[0]["sum_gray2"])) < 1e-5
    near = same.copy()
    near[1, h // 3, w // 2] ^= 0x40           # one pixel, all three channels
    near[2, h // 2:h // 2 + 2, 5:9] //= 2     # a 2 x 4 patch
    rec = engine.complexity(near[1:], prev0=near[0], mask=N.M_TEMPORAL_DCT, dct_mode=N.DCT_FULL)
    g = [co.bgr2gray(f).astype(np.float64) for f in near]
    for i in range(2):
        want = float(np.abs(scipy.fft.dctn(g[i], norm="ortho") - scipy.fft.dctn(g[i + 1], norm="ortho")).sum())
        assert want > 0 and _rel(rec[i]["temporal_dct_l1"], want) < RTOL, (i, float(rec[i]["temporal_dct_l1"]), want)


@pytest.mark.parametrize("h,w,rw,rh", [(270, 480, 64, 64), (64, 64, 64, 64), (90, 120, 40, 24),
                                       (270, 480, 480, 270), (300, 500, 200, 150), (540, 960, 960, 540),
                                       (300, 500, 262, 134), (300, 500, 256, 134), (300, 500, 134, 256), (300, 500, 128, 128),
                                       (300, 500, 486, 250), (540, 960, 946, 532)])
def test_dct_full_frame_parity_mode(engine, h, w, rw, rh):
    """config.json's own case: full-frame cv2.dct semantics on the resized plane (:363, :574-579).  Planes below 128 on a
    side: vector-ALU products; 480x270, 200x150, 960x540, 128x128, 486x250 (sides even and 2-3-5-smooth): FFT passes
    (radices 4, 2, 3, 5 all occur); 262x134, 256x134, 134x256, 946x532 (a side with another prime factor): MFMA products."""
    from rtvqa_amd import _native as N
    fr = _frames("natural", 3, h, w, seed=11)
    rec = engine.complexity(fr[1:], prev0=fr[0], mask=N.M_DCT | N.M_TEMPORAL_DCT, resize=(rw, rh),
                            dct_mode=N.DCT_FULL)
    for i in range(2):
        a = co.resize_linear(co.bgr2gray(fr[i + 1]), rw, rh)
        p = co.resize_linear(co.bgr2gray(fr[i]), rw, rh)
        assert _rel(rec[i]["dct_energy"], co.dct_energy_full(a)) < RTOL
        assert _rel(rec[i]["temporal_dct_l1"], co.temporal_dct_full(p, a)) < RTOL
    # AUTO picks FULL for planes up to 128x128, the 8x8 block metric above that
    rec2 = engine.complexity(fr[1:], prev0=fr[0], mask=N.M_TEMPORAL_DCT, resize=(rw, rh))
    if rw * rh <= 128 * 128:
        assert rec2[0]["temporal_dct_l1"] == rec[0]["temporal_dct_l1"]
    else:
        assert rec2[0]["temporal_dct_l1"] != rec[0]["temporal_dct_l1"]  # a different metric


# ---------------------------------------------------------------------------
def _canny_inputs():
    import scipy.ndimage as ndi
    out = {}
    out["noise"] = _rng(1).integers(0, 256, (97, 131), dtype=np.uint8)
    a = ndi.uniform_filter(_rng(2).integers(0, 256, (216, 336)).astype(float), 9)
    out["smooth"] = ((a - a.min()) / (a.max() - a.min()) * 255).astype(np.uint8)[8:-8, 8:-8]
    out["const"] = np.full((40, 70), 99, np.uint8)
    out["vstep"] = np.where(np.arange(200)[None, :] < 100, 40, 200).astype(np.uint8).repeat(70, 0)
    # a long thin weak ridge seeded by one strong pixel: hysteresis must cross many tiles
    r = np.full((80, 400), 20, np.uint8)
    r[40, :] = 50
    r[40, 5] = 255
    out["ridge"] = r
    yy, xx = np.mgrid[0:150, 0:260]
    out["spiral"] = ((np.sin(np.hypot(yy - 75, xx - 130) / 3.0) * 0.5 + 0.5) * 90 + 60).astype(np.uint8)
    return out


@pytest.mark.parametrize("name", ["noise", "smooth", "const", "vstep", "ridge", "spiral"])
@pytest.mark.parametrize("low,high", [(100, 200), (20, 60)])
def test_canny_count_and_map(engine, name, low, high):
    from rtvqa_amd import _native as N
    g = _canny_inputs()[name]
    h, w = g.shape
    fr = np.repeat(g[None, ..., None], 3, axis=3)  # B=G=R -> gray identity
    rec = engine.complexity(fr, mask=N.M_EDGE, canny=(low, high))
    cnt, strong, weak, edges = co.canny(g, low, high, want_map=True)
    assert (int(rec[0]["edge_strong"]), int(rec[0]["edge_weak"])) == (strong, weak)
    got = engine.debug_plane(2, 0, h, w)
    assert (got == edges).all(), "edge map differs at %d pixels" % int((got != edges).sum())
    assert int(rec[0]["edge_count"]) == cnt


@pytest.mark.parametrize("w", [1, 2, 3, 4, 5, 7, 8, 63, 64, 65, 255, 256, 257, 258, 259, 260, 261, 511, 512, 513, 514, 515, 516, 517, 770])
def test_canny_strip_and_border_widths(engine, w):
    """k_canny_nms3 works on 256-column strips with a one-column halo group and takes its interior (no realignment)
    path only when a strip's dword windows stay inside the image (x0 + 259 <= w): every width around those seams, the
    4-column minimum of the dword window (frames of 1-3 columns take the LDS-tile kernel k_canny_nms), two strips of rows, and
    thresholds that are negative / swapped / zero."""
    from rtvqa_amd import _native as N
    h = 67
    fr = _frames("natural", 2, h, w, seed=w)
    fr[1, :, : max(w // 3, 1)] //= 3  # a hard vertical step so narrow frames carry edges too
    for lo, hi in ((100, 200), (40, 20), (-5, 30), (0, 0)):
        rec = engine.complexity(fr, mask=N.M_EDGE, canny=(lo, hi))
        for i in range(2):
            g = co.bgr2gray(fr[i])
            cnt, strong, weak, emap = co.canny(g, min(lo, hi), max(lo, hi), want_map=True)
            assert (int(rec[i]["edge_count"]), int(rec[i]["edge_strong"]), int(rec[i]["edge_weak"])) == (cnt, strong, weak), (w, lo, hi, i)
            assert ((engine.debug_plane(2, i, h, w) != 0) == (emap != 0)).all(), (w, lo, hi, i)


def test_canny_batch_of_mixed_frames(engine):
    from rtvqa_amd import _native as N
    fr = np.concatenate([_frames("natural", 3, 200, 328, seed=3), _frames("noise", 2, 200, 328, seed=4)])
    rec = engine.complexity(fr, mask=N.M_EDGE)
    for i in range(5):
        assert int(rec[i]["edge_count"]) == co.canny(co.bgr2gray(fr[i]), 100, 200)[0], i


def test_canny_1080p_many_rounds(engine):
    """Low thresholds on natural content: long weak chains that cross many 64x64 tiles, so the
    hysteresis needs many list rounds (and the enqueue/dedup protocol is exercised for real)."""
    from rtvqa_amd import _native as N
    fr = _frames("natural", 4, 1080, 1920, seed=21)
    for low, high in ((30, 90), (100, 200)):
        rec = engine.complexity(fr, mask=N.M_EDGE, canny=(low, high))
        for i in range(4):
            cnt, strong, weak = co.canny(co.bgr2gray(fr[i]), low, high)
            assert (int(rec[i]["edge_strong"]), int(rec[i]["edge_weak"])) == (strong, weak)
            assert int(rec[i]["edge_count"]) == cnt, (low, high, i)


def test_canny_many_contents_one_batch(engine):
    """24 different 720p frames (natural + noise + mixed) in ONE batch at two threshold pairs: every frame's
    hysteresis runs through the wide rounds and the per-frame persistent tail concurrently with the others."""
    from rtvqa_amd import _native as N
    h, w = 720, 1280
    fr = np.concatenate([_frames("natural", 3, h, w, seed=100 + s) for s in range(6)] +
                        [_frames("noise", 2, h, w, seed=200)] +
                        [(_frames("natural", 2, h, w, seed=300) // 2 + _frames("noise", 2, h, w, seed=301) // 8)] +
                        [np.repeat(np.tile(np.arange(w, dtype=np.uint8)[None, :, None] // 3 * 3, (h, 1, 1)), 3, 2)[None]] +
                        [_frames("natural", 1, h, w, seed=400)])
    for low, high in ((100, 200), (25, 70)):
        rec = engine.complexity(fr, mask=N.M_EDGE, canny=(low, high))
        for i in range(fr.shape[0]):
            cnt, strong, weak = co.canny(co.bgr2gray(fr[i]), low, high)
            assert (int(rec[i]["edge_strong"]), int(rec[i]["edge_weak"]), int(rec[i]["edge_count"])) == \
                (strong, weak, cnt), (low, high, i)


# ---------------------------------------------------------------------------
@pytest.mark.parametrize("h,w", [(96, 128), (100, 200), (64, 80), (48, 48), (10, 300), (160, 272)])
@pytest.mark.parametrize("kind", ["natural", "noise"])
def test_block_sad(engine, h, w, kind):
    from rtvqa_amd import _native as N
    fr = _frames(kind, 3, h, w, seed=5)
    rec = engine.complexity(fr[1:], prev0=fr[0], mask=N.M_MOTION)
    for i in range(2):
        nb, sad, hist = co.block_sad(co.bgr2gray(fr[i]), co.bgr2gray(fr[i + 1]), 7)
        assert int(rec[i]["sad_blocks"]) == nb
        assert int(rec[i]["sad_sum"]) == sad
        assert (rec[i]["mv_d2_hist"] == hist).all()


@pytest.mark.parametrize("rng_", [0, 3, 7])
def test_block_sad_pan_and_range(engine, rng_):
    from rtvqa_amd import _native as N
    import scipy.ndimage as ndi
    a = ndi.uniform_filter(_rng(6).integers(0, 256, (200, 300)).astype(float), 7)
    base = ((a - a.min()) / (a.max() - a.min()) * 255).astype(np.uint8)
    prev = base[16:16 + 128, 16:16 + 192]
    curr = base[16 - 2:16 - 2 + 128, 16 + 3:16 + 3 + 192]
    f = lambda g: np.repeat(g[None, ..., None], 3, axis=3)
    rec = engine.complexity(f(curr), prev0=f(prev)[0], mask=N.M_MOTION, sad_range=rng_)
    nb, sad, hist = co.block_sad(prev, curr, rng_)
    assert int(rec[0]["sad_sum"]) == sad and (rec[0]["mv_d2_hist"] == hist).all()
    if rng_ >= 3:
        assert hist[13] >= (128 // 16 - 2) * (192 // 16 - 2)  # interior blocks find (2,-3) exactly
    # identical frames: all-zero vectors, zero SAD
    rec = engine.complexity(f(prev), prev0=f(prev)[0], mask=N.M_MOTION, sad_range=rng_)
    assert rec[0]["sad_sum"] == 0 and rec[0]["mv_d2_hist"][0] == rec[0]["sad_blocks"] == nb


# ---------------------------------------------------------------------------
def _check_quality(engine, ref, dist, planes, mode_name):
    from rtvqa_amd import _native as N
    mode = N.SSIM_GAUSS if mode_name == "gauss" else N.SSIM_FFMPEG
    res = engine.quality(ref, dist, planes, mode)
    for i in range(ref.shape[0]):
        sse, ssim = pl.frame_quality(ref[i], dist[i], planes, mode_name)
        for p in range(len(planes)):
            assert int(res[i, p]["sse"]) == sse[p], (i, p)
            assert _rel(res[i, p]["ssim"], ssim[p]) < RTOL, (i, p, res[i, p]["ssim"], ssim[p])
    return res


@pytest.mark.parametrize("mode", ["gauss", "ffmpeg"])
@pytest.mark.parametrize("h,w", [(24, 40), (11, 11), (140, 530), (150, 1030), (67, 259)])
def test_quality_gray_planes(engine, mode, h, w):
    from rtvqa_amd.engine import gray_planes
    from rtvqa_amd import synth
    ref = np.ascontiguousarray(_frames("natural", 2, h, w, seed=8)[..., 0])
    dist = synth.distort(ref)
    _check_quality(engine, ref, dist, gray_planes(h, w), mode)


@pytest.mark.parametrize("mode", ["gauss", "ffmpeg"])
def test_quality_bgr_and_yuv420p(engine, mode):
    from rtvqa_amd.engine import bgr_planes, yuv420p_planes
    from rtvqa_amd import synth
    h, w = 72, 104
    ref = _frames("natural", 3, h, w, seed=9)
    dist = synth.distort(ref)
    _check_quality(engine, ref, dist, bgr_planes(h, w), mode)
    yuv_r = _rng(10).integers(0, 256, (2, h * w * 3 // 2), dtype=np.uint8)
    yuv_d = np.clip(yuv_r.astype(int) + _rng(11).integers(-5, 6, yuv_r.shape), 0, 255).astype(np.uint8)
    _check_quality(engine, yuv_r, yuv_d, yuv420p_planes(h, w), mode)


def test_quality_known_answers(engine):
    from rtvqa_amd import _native as N
    from rtvqa_amd.engine import gray_planes
    ref = _rng(12).integers(0, 255, (2, 64, 96), dtype=np.uint8)
    for mode in (N.SSIM_GAUSS, N.SSIM_FFMPEG):
        res = engine.quality(ref, ref, gray_planes(64, 96), mode)
        assert (res["sse"] == 0).all() and np.allclose(res["ssim"], 1.0, atol=1e-6)
        res = engine.quality(ref, ref + 1, gray_planes(64, 96), mode)
        assert (res["sse"] == 64 * 96).all()  # MSE 1 -> 48.1308 dB


# ---------------------------------------------------------------------------
def test_device_resident_equals_host_and_batching(engine):
    from rtvqa_amd import _native as N
    fr = _frames("natural", 6, 96, 160, seed=13)
    host = engine.complexity(fr[1:], prev0=fr[0], mask=N.M_ALL, dct_mode=N.DCT_BLOCK8)
    dev = engine.upload(fr)
    devr = engine.complexity(dev.slice(1, 6), prev0=dev.frame(0), mask=N.M_ALL, dct_mode=N.DCT_BLOCK8)
    names = [n for n in host.dtype.names if n != "hyst_steps"]  # (a diagnostic that depends on scheduling)
    for name in names:
        assert (host[name] == devr[name]).all(), name
    # one frame at a time, chained through prev0, gives the same records as the batch
    for i in range(5):
        one = engine.complexity(fr[i + 1:i + 2], prev0=fr[i], mask=N.M_ALL, dct_mode=N.DCT_BLOCK8)
        for name in names:
            assert (one[0][name] == host[i][name]).all(), (i, name)


def test_torch_tensor_frames_zero_copy(engine):
    """PyTorch-ROCm is plumbing: a CUDA(HIP) uint8 tensor's data_ptr() goes straight into the C ABI
    (one HIP runtime in the process), and gives the same records as host frames."""
    import torch
    from rtvqa_amd import _native as N
    from rtvqa_amd.engine import DeviceFrames, bgr_planes
    from rtvqa_amd import synth
    fr = _frames("natural", 4, 96, 160, seed=17)
    dist = synth.distort(fr)
    t_ref, t_dist = torch.from_numpy(fr).cuda(), torch.from_numpy(dist).cuda()
    torch.cuda.synchronize()
    d_ref, d_dist = DeviceFrames.from_torch(t_ref), DeviceFrames.from_torch(t_dist)
    host = engine.complexity(dist[1:], prev0=dist[0], mask=N.M_ALL, dct_mode=N.DCT_BLOCK8)
    dev = engine.complexity(d_dist.slice(1, 4), prev0=d_dist.frame(0), mask=N.M_ALL, dct_mode=N.DCT_BLOCK8)
    for name in host.dtype.names:
        assert (host[name] == dev[name]).all(), name
    qh = engine.quality(fr, dist, bgr_planes(96, 160), N.SSIM_GAUSS)
    qd = engine.quality(d_ref, d_dist, bgr_planes(96, 160), N.SSIM_GAUSS)
    assert (qh["sse"] == qd["sse"]).all() and (qh["ssim"] == qd["ssim"]).all()


def test_full_size_1080p_parity_and_properties(engine):
    """BASELINE.json's frame size: two 1080p frames against the oracle, plus size-independent
    properties (Parseval, bin totals, identical-pair invariants)."""
    from rtvqa_amd import _native as N
    from rtvqa_amd import synth
    from rtvqa_amd.engine import bgr_planes
    h, w = 1080, 1920
    fr = _frames("natural", 3, h, w, seed=14)
    rec = engine.complexity(fr[1:], prev0=fr[0], mask=N.M_ALL, dct_mode=N.DCT_BLOCK8)
    for i in range(2):
        g, gp = co.bgr2gray(fr[i + 1]), co.bgr2gray(fr[i])
        assert (rec[i]["hist_gray"] == co.hist_u8(g)).all()
        assert int(rec[i]["hist_bgr"].sum()) == 3 * h * w
        assert int(rec[i]["sum_gray2"]) == int((g.astype(np.int64) ** 2).sum())
        assert _rel(rec[i]["dct_energy"], float(rec[i]["sum_gray2"])) < RTOL
        e, l1, _ = co.dct8x8(gp, g)
        assert _rel(rec[i]["temporal_dct_l1"], l1) < RTOL
        assert int(rec[i]["edge_count"]) == co.canny(g, 100, 200)[0]
        nb, sad, hist = co.block_sad(gp, g, 7)
        assert int(rec[i]["sad_sum"]) == sad and (rec[i]["mv_d2_hist"] == hist).all() and rec[i]["sad_blocks"] == nb
    dist = synth.distort(fr[:2])
    _check_quality(engine, fr[:2], dist, bgr_planes(h, w), "gauss")
    _check_quality(engine, fr[:2], dist, bgr_planes(h, w), "ffmpeg")
    same = engine.quality(fr[:1], fr[:1], bgr_planes(h, w), N.SSIM_GAUSS)
    assert (same["sse"] == 0).all() and np.allclose(same["ssim"], 1.0, atol=1e-6)


def test_full_size_2160p_parity(engine):
    """BASELINE.json configs[3]'s frame size (3840x2160), one frame pair against the oracle."""
    from rtvqa_amd import _native as N
    from rtvqa_amd import synth
    from rtvqa_amd.engine import gray_planes
    h, w = 2160, 3840
    fr = _frames("natural", 2, h, w, seed=15)
    rec = engine.complexity(fr[1:], prev0=fr[0], mask=N.M_ALL, dct_mode=N.DCT_BLOCK8)
    g, gp = co.bgr2gray(fr[1]), co.bgr2gray(fr[0])
    assert (rec[0]["hist_gray"] == co.hist_u8(g)).all()
    assert (rec[0]["hist_bgr"][2] == co.hist_u8(fr[1], offset=2, step=3)).all()
    assert int(rec[0]["edge_count"]) == co.canny(g, 100, 200)[0]
    nb, sad, hist = co.block_sad(gp, g, 7)
    assert int(rec[0]["sad_sum"]) == sad and (rec[0]["mv_d2_hist"] == hist).all() and rec[0]["sad_blocks"] == nb
    e, l1, _ = co.dct8x8(gp, g)
    assert _rel(rec[0]["dct_energy"], e) < RTOL and _rel(rec[0]["temporal_dct_l1"], l1) < RTOL
    gd = synth.distort(g[None])
    _check_quality(engine, g[None], gd, gray_planes(h, w), "gauss")
    _check_quality(engine, g[None], gd, gray_planes(h, w), "ffmpeg")


@pytest.mark.parametrize("seed", range(10))
def test_random_geometries_all_kernels(engine, seed):
    """Ragged, odd and tiny frame geometries drawn at random: every kernel against the oracle."""
    from rtvqa_amd import _native as N
    from rtvqa_amd import synth
    from rtvqa_amd.engine import bgr_planes
    r = _rng(1000 + seed)
    h, w = int(r.integers(16, 180)), int(r.integers(16, 300))
    kind = "natural" if seed % 2 else "noise"
    fr = _frames(kind, 3, h, w, seed=seed)
    lo, hi = (100, 200) if seed % 3 else (15, 45)
    rec = engine.complexity(fr[1:], prev0=fr[0], mask=N.M_ALL, dct_mode=N.DCT_BLOCK8, canny=(lo, hi),
                            sad_range=int(r.integers(0, 8)))
    for i in range(2):
        g, gp = co.bgr2gray(fr[i + 1]), co.bgr2gray(fr[i])
        assert (rec[i]["hist_gray"] == co.hist_u8(g)).all(), (h, w)
        for c in range(3):
            assert (rec[i]["hist_bgr"][c] == co.hist_u8(fr[i + 1], offset=c, step=3)).all(), (h, w)
        e, l1, _ = co.dct8x8(gp, g)
        assert _rel(rec[i]["dct_energy"], e) < RTOL and _rel(rec[i]["temporal_dct_l1"], max(l1, 1e-9)) < RTOL or l1 == 0
        cnt, strong, weak = co.canny(g, lo, hi)
        assert (int(rec[i]["edge_strong"]), int(rec[i]["edge_weak"]), int(rec[i]["edge_count"])) == (strong, weak, cnt), (h, w)
    dist = synth.distort(fr)
    if h >= 11 and w >= 11:
        _check_quality(engine, fr, dist, bgr_planes(h, w), "gauss")
    if h >= 8 and w >= 8:
        _check_quality(engine, fr, dist, bgr_planes(h, w), "ffmpeg")


@pytest.mark.parametrize("h,w", [(1, 1), (1, 9), (9, 1), (2, 2), (3, 5), (7, 7), (9, 9), (15, 17), (1, 300), (300, 1),
                                 (2, 129), (5, 64)])
def test_tiny_frames_all_complexity_kernels(engine, h, w):
    """Frames smaller than a DCT block, a SAD block, a Sobel window or a 4-pixel lane group."""
    from rtvqa_amd import _native as N
    fr = _rng(h * 1000 + w).integers(0, 256, (3, h, w, 3), dtype=np.uint8)
    fr[2, :, : (w + 1) // 2] = 255  # a hard step so tiny frames still carry edges
    fr[2, :, (w + 1) // 2:] = 0
    rec = engine.complexity(fr[1:], prev0=fr[0], mask=N.M_ALL, dct_mode=N.DCT_BLOCK8, canny=(20, 60))
    for i in range(2):
        g, gp = co.bgr2gray(fr[i + 1]), co.bgr2gray(fr[i])
        assert (rec[i]["hist_gray"] == co.hist_u8(g)).all()
        for c in range(3):
            assert (rec[i]["hist_bgr"][c] == co.hist_u8(fr[i + 1], offset=c, step=3)).all()
        assert int(rec[i]["sum_gray2"]) == int((g.astype(np.int64) ** 2).sum())
        e, l1, _ = co.dct8x8(gp, g)
        assert e == 0 or _rel(rec[i]["dct_energy"], e) < RTOL
        assert l1 == 0 or _rel(rec[i]["temporal_dct_l1"], l1) < RTOL
        cnt, strong, weak, emap = co.canny(g, 20, 60, want_map=True)
        assert (int(rec[i]["edge_strong"]), int(rec[i]["edge_weak"]), int(rec[i]["edge_count"])) == (strong, weak, cnt)
        assert ((engine.debug_plane(2, i, h, w) != 0) == (emap != 0)).all()
        nb, sad, hist = co.block_sad(gp, g, 7)
        assert (int(rec[i]["sad_sum"]), int(rec[i]["sad_blocks"])) == (sad, nb)
        assert (rec[i]["mv_d2_hist"] == hist).all()
    # full-frame DCT mode and the resize path on the same tiny frames
    rec = engine.complexity(fr[1:], prev0=fr[0], mask=N.M_DCT | N.M_TEMPORAL_DCT | N.M_GRAY_HIST, dct_mode=N.DCT_FULL,
                            resize=(6, 4))
    for i in range(2):
        p_, a_ = (co.resize_linear(co.bgr2gray(fr[i + k]), 6, 4) for k in (0, 1))
        e, l1 = co.dct_energy_full(a_), co.temporal_dct_full(p_, a_)
        assert e == 0 or _rel(rec[i]["dct_energy"], e) < RTOL
        assert l1 == 0 or _rel(rec[i]["temporal_dct_l1"], l1) < RTOL
        assert (rec[i]["hist_gray"] == co.hist_u8(co.bgr2gray(co.resize_linear(fr[i + 1], 6, 4)))).all()


@pytest.mark.parametrize("seed", range(6))
def test_random_geometries_sad_and_resize(engine, seed):
    from rtvqa_amd import _native as N
    r = _rng(2000 + seed)
    h, w = int(r.integers(16, 200)), int(r.integers(16, 330))
    rng_ = int(r.integers(0, 8))
    fr = _frames("natural", 3, h, w, seed=50 + seed)
    rec = engine.complexity(fr[1:], prev0=fr[0], mask=N.M_MOTION, sad_range=rng_)
    for i in range(2):
        nb, sad, hist = co.block_sad(co.bgr2gray(fr[i]), co.bgr2gray(fr[i + 1]), rng_)
        assert int(rec[i]["sad_blocks"]) == nb and int(rec[i]["sad_sum"]) == sad and (rec[i]["mv_d2_hist"] == hist).all(), (h, w, rng_)
    rw, rh = int(r.integers(8, 200)), int(r.integers(8, 150))
    rec = engine.complexity(fr[1:], prev0=fr[0], mask=N.M_GRAY_HIST | N.M_COLOR_HIST | N.M_EDGE | N.M_DCT | N.M_TEMPORAL_DCT,
                            resize=(rw, rh), dct_mode=N.DCT_BLOCK8)
    for i in range(2):
        a = co.resize_linear(co.bgr2gray(fr[i + 1]), rw, rh)
        p = co.resize_linear(co.bgr2gray(fr[i]), rw, rh)
        rb = co.resize_linear(fr[i + 1], rw, rh)
        b = co.bgr2gray(rb)
        assert (engine.debug_plane(0, i, rh, rw) == a).all() and (engine.debug_plane(1, i, rh, rw) == b).all(), (h, w, rw, rh)
        assert (rec[i]["hist_gray"] == co.hist_u8(b)).all()
        assert int(rec[i]["edge_count"]) == co.canny(b, 100, 200)[0]
        e, l1, _ = co.dct8x8(p, a)
        assert _rel(rec[i]["dct_energy"], e) < RTOL and (l1 == 0 or _rel(rec[i]["temporal_dct_l1"], l1) < RTOL)


def _orb_oracle(frame):
    return co.orb64_count(co.bgr2gray(co.resize_linear(frame, 64, 64)))


@pytest.mark.parametrize("h,w", [(64, 64), (128, 128), (270, 480), (97, 131), (40, 52), (1080, 1920)])
def test_orb_keypoint_count(engine, h, w):
    """process_orb_frame_for_parallel (:367-389): FAST-9/16 + NMS at the centre of the 64x64 thumbnail."""
    from rtvqa_amd import _native as N
    rng = _rng(h + w)
    n = 48 if h < 1000 else 6
    fr = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8) if h < 1000 else _frames("natural", n, h, w, seed=3)
    # plant isolated bright / dark spots over a flat centre in some frames so that keypoints do occur
    cy, cx = h // 2, w // 2
    ry, rx = max(4 * h // 64, 4), max(4 * w // 64, 4)
    for i in range(0, n, 3):
        fr[i, cy - ry:cy + ry, cx - rx:cx + rx] = 60 if i % 2 else 200
        oy, ox = int(rng.integers(-h // 64 - 1, h // 64 + 2)), int(rng.integers(-w // 64 - 1, w // 64 + 2))
        sy, sx = max(h // 64, 1), max(w // 64, 1)
        fr[i, cy + oy - sy:cy + oy + sy, cx + ox - sx:cx + ox + sx] = 255 if i % 2 else 0
    rec = engine.complexity(fr, mask=N.M_ORB)
    want = [_orb_oracle(f) for f in fr]
    got = [(int(r["orb_keypoints"]), int(r["orb_response"])) for r in rec]
    assert got == want
    assert {c for c, _ in want} == {0, 1} or h >= 1000, "test frames should exercise both outcomes"
    # the metric ignores the configured resize and rides along with the rest of the suite unchanged
    rec2 = engine.complexity(fr[:4], mask=N.M_ALL, resize=(48, 32))
    assert [(int(r["orb_keypoints"]), int(r["orb_response"])) for r in rec2] == want[:4]
    dev = engine.upload(fr[:4])
    rec3 = engine.complexity(dev, mask=N.M_ORB | N.M_GRAY_HIST)
    assert [(int(r["orb_keypoints"]), int(r["orb_response"])) for r in rec3] == want[:4]


@pytest.mark.parametrize("h,w,kind", [(97, 131, "natural"), (64, 64, "natural"), (50, 300, "noise"), (33, 40, "natural"),
                                      (135, 240, "natural"), (270, 480, "natural"), (16, 16, "noise"), (40, 900, "natural"),
                                      (1, 1, "noise"), (5, 7, "noise"), (9, 9, "noise"), (2, 40, "noise"), (31, 64, "noise")])
def test_farneback_motion_parity(engine, h, w, kind):
    """VQA_MOTION_FARNEBACK (the reference's own motion metric, :340-343) against the oracle restatement.
    The kernels evaluate the oracle's float/double expressions in its order with contraction off, so the
    mean magnitude agrees far inside the 1e-4 bar; the batch also crosses a chunk of several pairs."""
    from rtvqa_amd import _native as N
    fr = _frames(kind, 4, h, w, seed=h + w)
    rec = engine.complexity(fr[1:], prev0=fr[0], mask=N.M_MOTION, motion_mode=N.MOTION_FARNEBACK)
    gray = [co.bgr2gray(f) for f in fr]
    for i in range(3):
        want = co.farneback(gray[i], gray[i + 1])
        got = float(rec[i]["flow_mag_mean"])
        assert abs(got - want) <= RTOL * want + 1e-7, (i, got, want)
        assert int(rec[i]["sad_blocks"]) == 0  # the SAD kernel did not run
    # no previous frame for the first one: 0.0, as the reference's None check (:324-325)
    rec = engine.complexity(fr[1:3], mask=N.M_MOTION, motion_mode=N.MOTION_FARNEBACK)
    assert float(rec[0]["flow_mag_mean"]) == 0.0 and int(rec[0]["has_prev"]) == 0
    assert abs(float(rec[1]["flow_mag_mean"]) - co.farneback(gray[1], gray[2])) <= RTOL * co.farneback(gray[1], gray[2]) + 1e-7


def _fuzz_case_frames(case):
    """The frames scripts/fuzz_parity.py draws for a case number (same generator calls, same order)."""
    from rtvqa_amd import synth
    r = _rng(case)
    h, w = int(r.integers(1, 200)), int(r.integers(1, 320))
    kind, n = int(r.integers(0, 3)), int(r.integers(1, 4))
    if kind == 0:
        return r.integers(0, 256, (n + 1, h, w, 3), dtype=np.uint8)
    if kind == 1:
        return synth.s_natural(n + 1, h, w, seed=case)
    return np.repeat(r.integers(0, 256, (n + 1, (h + 7) // 8, (w + 7) // 8, 3), dtype=np.uint8), 8, axis=1).repeat(8, axis=2)[:, :h, :w]


@pytest.mark.parametrize("case", [40610, 41571])
def test_farneback_border_discontinuity_cases(engine, case):
    """The two round-2 fuzz outliers (35x31 uniform noise, 129x34 blocky noise), kept as tests with their cause.
    FarnebackUpdateMatrices warps R1 by the current flow and DROPS the warped term when the source cell leaves the
    frame (y1 >= 0 && y1 < h-1 ...).  On these frames a pixel of the top border row holds, after the second
    iteration, a vertical flow of +5.8e-8 in the float oracle (in frame) and of about -1e-7 in any evaluation whose
    double box sums round differently (out of frame): the hard test flips, that pixel's matrix changes by O(1), and
    on a frame this small its 15x15 neighbourhood moves the MEAN magnitude by 4.1e-4.  It is a discontinuity of the
    published algorithm, not of a summation order: the tap-order and the OpenCV-order (sliding, float-difference)
    box sums of the C oracle agree to 2e-8 here, and the float64 restatement sits on the other side with the device.
    Bar of the mode (include/vqa.h, flow_mag_mean): 1e-4 relative, except on frames where a border pixel's flow is
    within rounding of that test, where only the looser 2e-3 holds and both sides are valid evaluations."""
    from rtvqa_amd import _native as N
    fr = _fuzz_case_frames(case)
    rec = engine.complexity(fr[1:2], prev0=fr[0], mask=N.M_MOTION, motion_mode=N.MOTION_FARNEBACK)
    g, gp = co.bgr2gray(fr[1]), co.bgr2gray(fr[0])
    got = float(rec[0]["flow_mag_mean"])
    c_val, np_val = co.farneback(gp, g), no.farneback_mean_mag(gp, g)
    assert 1e-4 < abs(c_val - np_val) / np_val < 2e-3            # the two restatements straddle the discontinuity
    assert min(abs(got - c_val) / c_val, abs(got - np_val) / np_val) <= RTOL   # the device is one of the two evaluations
    assert abs(got - c_val) / c_val < 2e-3 and abs(got - np_val) / np_val < 2e-3


def test_farneback_known_translation_1080p(engine):
    """Full size: a smooth texture panned by (2, 1) pixels per frame -> mean |flow| = sqrt(5) within 1 %;
    identical frames -> ~0; and the suite's other metrics are untouched by the motion mode."""
    import scipy.ndimage as ndi
    from rtvqa_amd import _native as N
    a = ndi.gaussian_filter(_rng(77).integers(0, 256, (1080 + 16, 1920 + 16)).astype(np.float32), 2.5)
    a = ((a - a.min()) / (a.max() - a.min()) * 255).astype(np.uint8)
    fr = np.stack([np.repeat(a[8 + k:8 + k + 1080, 8 + 2 * k:8 + 2 * k + 1920, None], 3, axis=2) for k in range(3)] +
                  [np.repeat(a[10:1090, 12:1932, None], 3, axis=2)])  # frame 3 == frame 2
    rec = engine.complexity(fr[1:], prev0=fr[0], mask=N.M_ALL, dct_mode=N.DCT_BLOCK8, motion_mode=N.MOTION_FARNEBACK)
    for i in range(2):
        assert abs(float(rec[i]["flow_mag_mean"]) - 5 ** 0.5) < 0.01 * 5 ** 0.5
    assert float(rec[2]["flow_mag_mean"]) < 1e-3
    ref = engine.complexity(fr[1:], prev0=fr[0], mask=N.M_ALL & ~N.M_MOTION, dct_mode=N.DCT_BLOCK8)
    for k in ("hist_gray", "edge_count", "dct_energy", "temporal_dct_l1", "orb_keypoints"):
        assert (rec[k] == ref[k]).all(), k


def test_farneback_does_not_depend_on_the_batch_it_rides_in(engine):
    """The fused flow iteration picks its row-strip count from the batch (residency rounds x rows marched): a pair submitted
    alone marches many short strips, the same pair inside a 40-pair batch a few long ones.  Since round 6 the running column
    sums restart at every multiple of 16 rows whether or not a strip begins there, strips begin only at such rows, and the
    magnitudes are summed in fixed point: the metric is the SAME BITS in any batch (rounds 4-5: 1e-6 apart), at every
    geometry class - one strip, several strips, a height that is no multiple of 16, a frame of a few restart periods."""
    from rtvqa_amd import _native as N
    for h, w, seed in ((540, 960, 31), (1080, 1920, 32), (200, 300, 33), (50, 70, 34)):
        fr = _frames("natural", 2, h, w, seed=seed)
        alone = engine.complexity(fr[1:2], prev0=fr[0], mask=N.M_MOTION, motion_mode=N.MOTION_FARNEBACK)
        # 40 pairs: even slots hold frame 0, odd slots frame 1 -> every pair is (0 -> 1) or (1 -> 0)
        n = 41 if h < 1080 else 13
        rep = np.stack([fr[i & 1] for i in range(n)])
        many = engine.complexity(rep[1:], prev0=rep[0], mask=N.M_MOTION, motion_mode=N.MOTION_FARNEBACK)
        few = engine.complexity(rep[1:4], prev0=rep[0], mask=N.M_MOTION, motion_mode=N.MOTION_FARNEBACK)
        a = float(alone[0]["flow_mag_mean"])
        fwd = many["flow_mag_mean"][0::2]
        assert a > 0 and (fwd == fwd[0]).all() and (many["flow_mag_mean"][1::2] == many["flow_mag_mean"][1]).all()
        assert float(fwd[0]) == a == float(few[0]["flow_mag_mean"]), (h, w, float(fwd[0]), a)
        assert float(few[1]["flow_mag_mean"]) == float(many["flow_mag_mean"][1])


def test_many_small_frames_one_batch(engine):
    """Thousands of frames in one submit (frames ride in gridDim.y; per-frame lists, counters, partials)."""
    from rtvqa_amd import _native as N
    n, h, w = 3000, 24, 40
    fr = _rng(8).integers(0, 256, (n + 1, h, w, 3), dtype=np.uint8)
    fr[::7, 8:16, 10:30] = 255
    rec = engine.complexity(fr[1:], prev0=fr[0], mask=N.M_ALL, dct_mode=N.DCT_BLOCK8, canny=(30, 90))
    for i in list(range(0, n, 211)) + [n - 1]:
        g, gp = co.bgr2gray(fr[i + 1]), co.bgr2gray(fr[i])
        assert (rec[i]["hist_gray"] == co.hist_u8(g)).all()
        assert int(rec[i]["edge_count"]) == co.canny(g, 30, 90)[0]
        nb, sad, hist = co.block_sad(gp, g, 7)
        assert (int(rec[i]["sad_sum"]), int(rec[i]["sad_blocks"])) == (sad, nb) and (rec[i]["mv_d2_hist"] == hist).all()
        e, l1, _ = co.dct8x8(gp, g)
        assert _rel(rec[i]["dct_energy"], e) < RTOL and _rel(rec[i]["temporal_dct_l1"], l1) < RTOL
        assert int(rec[i]["orb_keypoints"]) == co.orb64_count(co.bgr2gray(co.resize_linear(fr[i + 1], 64, 64)))[0]
    from rtvqa_amd.engine import gray_planes
    q = engine.quality(fr[:-1, ..., 1], fr[1:, ..., 1], gray_planes(h, w), N.SSIM_GAUSS)
    for i in (0, 1499, n - 1):
        a, b = np.ascontiguousarray(fr[i, ..., 1]), np.ascontiguousarray(fr[i + 1, ..., 1])
        assert int(q[i, 0]["sse"]) == co.sse_plane(a, b) and _rel(q[i, 0]["ssim"], co.ssim_gauss(a, b)) < RTOL


def test_more_than_65535_frames_in_one_submit(engine):
    """Frames ride in gridDim.y (<= 65535); the C ABI slices larger batches internally, carrying the previous frame
    across the slice seam (32768 | 32769)."""
    from rtvqa_amd import _native as N
    from rtvqa_amd.engine import gray_planes
    n, h, w = 70000, 16, 24
    fr = _rng(81).integers(0, 256, (n + 1, h, w, 3), dtype=np.uint8)
    fr[::5, 4:12, 6:18] = 255
    rec = engine.complexity(fr[1:], prev0=fr[0], mask=N.M_ALL, dct_mode=N.DCT_BLOCK8, canny=(30, 90))
    assert rec.shape == (n,) and (rec["has_prev"] == 1).all()
    for i in [0, 1, 32766, 32767, 32768, 32769, 65534, 65535, 65536, n - 1]:
        g, gp = co.bgr2gray(fr[i + 1]), co.bgr2gray(fr[i])
        assert (rec[i]["hist_gray"] == co.hist_u8(g)).all(), i
        assert int(rec[i]["edge_count"]) == co.canny(g, 30, 90)[0], i
        nb, sad, hist = co.block_sad(gp, g, 7)
        assert (int(rec[i]["sad_sum"]), int(rec[i]["sad_blocks"])) == (sad, nb) and (rec[i]["mv_d2_hist"] == hist).all(), i
        e, l1, _ = co.dct8x8(gp, g)
        assert _rel(rec[i]["dct_energy"], e) < RTOL and _rel(rec[i]["temporal_dct_l1"], l1) < RTOL, i
    q = engine.quality(fr[:-1, ..., 1], fr[1:, ..., 1], gray_planes(h, w), N.SSIM_GAUSS)
    assert q.shape[0] == n
    for i in (0, 32767, 32768, 65536, n - 1):
        a, b = np.ascontiguousarray(fr[i, ..., 1]), np.ascontiguousarray(fr[i + 1, ..., 1])
        assert int(q[i, 0]["sse"]) == co.sse_plane(a, b) and _rel(q[i, 0]["ssim"], co.ssim_gauss(a, b)) < RTOL, i
    q = engine.quality(fr[:-1, ..., 1], fr[1:, ..., 1], gray_planes(h, w), N.SSIM_FFMPEG)
    for i in (0, 32768, n - 1):
        a, b = np.ascontiguousarray(fr[i, ..., 1]), np.ascontiguousarray(fr[i + 1, ..., 1])
        assert int(q[i, 0]["sse"]) == co.sse_plane(a, b) and _rel(q[i, 0]["ssim"], co.ssim_ffmpeg(a, b)) < RTOL, i


def test_pruned_block_sad_variant_is_bit_identical():
    """Lab build, VQA_SAD_VARIANT=2 (successive-elimination search, k_block_sad_sea) must return the exhaustive winner: sad_sum
    and the d^2 histogram bit for bit, on natural, noise, ragged and tiny planes and for every search range."""
    import subprocess
    import sys
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r)\n"
        "import rtvqa_amd\n"
        "from rtvqa_amd import _native as N, synth\n"
        "from oracle import c_oracle as co\n"
        "eng = rtvqa_amd.Engine(0)\n"
        "cases = [('natural', 270, 480, 7), ('noise', 97, 131, 7), ('natural', 64, 200, 3), ('noise', 33, 47, 0), ('natural', 1080, 1920, 7), ('noise', 16, 16, 5)]\n"
        "for kind, h, w, R in cases:\n"
        "    fr = synth.s_natural(3, h, w, seed=5) if kind == 'natural' else synth.s_noise(3, h, w, seed=5)\n"
        "    rec = eng.complexity(fr[1:], prev0=fr[0], mask=N.M_MOTION, sad_range=R)\n"
        "    for i in range(2):\n"
        "        nb, sad, hist = co.block_sad(co.bgr2gray(fr[i]), co.bgr2gray(fr[i + 1]), R)\n"
        "        assert int(rec[i]['sad_blocks']) == nb and int(rec[i]['sad_sum']) == sad and (rec[i]['mv_d2_hist'] == hist).all(), (kind, h, w, R, i)\n"
        "print('PRUNED-OK')\n" % REPO_ROOT
    )
    from rtvqa_amd import _native as N
    env = dict(os.environ, VQA_SAD_VARIANT="2", VQA_LIB_PATH=N.LAB_LIB_PATH)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600, cwd=REPO_ROOT)
    assert r.returncode == 0 and "PRUNED-OK" in r.stdout, (r.stdout[-400:], r.stderr[-1200:])


_VARIANT_CODE = (
    "import sys, numpy as np; sys.path.insert(0, %r)\n"
    "import rtvqa_amd\n"
    "from rtvqa_amd import _native as N, synth\n"
    "from rtvqa_amd.engine import bgr_planes\n"
    "from oracle import c_oracle as co\n"
    "from oracle import pipeline as pl\n"
    "eng = rtvqa_amd.Engine(0)\n"
    "assert eng.lib.vqa_build_flavour() == @FLAVOUR@, eng.lib.vqa_build_flavour()\n"
    "eng.set_option(N.OPT_HYST_STATS, @STATS@)\n"
    "for kind, h, w in (('natural', 270, 480), ('noise', 97, 131), ('natural', 64, 200)):\n"
    "    fr = synth.s_natural(4, h, w, seed=9) if kind == 'natural' else synth.s_noise(4, h, w, seed=9)\n"
    "    rec = eng.complexity(fr[1:], prev0=fr[0], mask=N.M_ALL, dct_mode=N.DCT_BLOCK8, canny=(40, 120))\n"
    "    for i in range(3):\n"
    "        g, gp = co.bgr2gray(fr[i + 1]), co.bgr2gray(fr[i])\n"
    "        e, l1, _ = co.dct8x8(gp, g)\n"
    "        assert abs(rec[i]['dct_energy'] - e) <= 1e-4 * e and abs(rec[i]['temporal_dct_l1'] - l1) <= 1e-4 * l1, ('dct', kind, i)\n"
    "        assert int(rec[i]['edge_count']) == co.canny(g, 40, 120)[0] and not rec[i]['hyst_overflow'], ('canny', kind, i)\n"
    "        assert (int(rec[i]['hyst_steps']) > 0) == bool(@STATS@), ('hyst_steps', kind, i)\n"
    "        nb, sad, hist = co.block_sad(gp, g, 7)\n"
    "        assert int(rec[i]['sad_sum']) == sad and (rec[i]['mv_d2_hist'] == hist).all(), ('sad', kind, i)\n"
    "    q = eng.quality(fr[:2], fr[1:3], bgr_planes(h, w), N.SSIM_GAUSS)\n"
    "    sse, ssim = pl.frame_quality(fr[0], fr[1], bgr_planes(h, w), 'gauss')\n"
    "    for p in range(3):\n"
    "        assert int(q[0, p]['sse']) == sse[p] and abs(q[0, p]['ssim'] - ssim[p]) <= 1e-4 * abs(ssim[p]), ('ssim', kind, p)\n"
    "    fb = eng.complexity(fr[1:3], prev0=fr[0], mask=N.M_MOTION, motion_mode=N.MOTION_FARNEBACK)\n"
    "    for i in range(2):\n"
    "        want = co.farneback(co.bgr2gray(fr[i]), co.bgr2gray(fr[i + 1]))\n"
    "        assert abs(float(fb[i]['flow_mag_mean']) - want) <= 1e-4 * want + 1e-7, ('farneback', kind, i)\n"
    "print('VARIANT-OK')\n"
)


@pytest.mark.parametrize("knob", ["VQA_DCT_VARIANT=4", "VQA_DCT_VARIANT=3", "VQA_DCT_VARIANT=1", "VQA_DCT_LOAD_EARLY=1", "VQA_DCT_FCH=3",
                                  "VQA_SSIM_VARIANT=1", "VQA_SSIM_VARIANT=4", "VQA_NMS_VARIANT=1", "VQA_NMS_VARIANT=2", "VQA_HYST_SUB=1",
                                  "VQA_HYST_SUB=5", "VQA_HYST_WIDE=1", "VQA_FB_VARIANT=1", "VQA_FB_LEVEL_VARIANT=1", "NONE=0"])
def test_ab_knob_variants_keep_parity(knob):
    """LAB build (csrc/lab/libvqa_hip_lab.so, loaded through VQA_LIB_PATH): every superseded kernel variant kept for
    re-measurement (LAB_NOTES.md) still matches the oracle (selectors are read once per process => one subprocess per
    setting; NONE=0 is the lab build with every selector at its shipped default)."""
    import subprocess
    import sys
    from rtvqa_amd import _native as N
    k, v = knob.split("=")
    env = dict(os.environ, VQA_LIB_PATH=N.LAB_LIB_PATH)
    env[k] = v
    code = _VARIANT_CODE.replace("@FLAVOUR@", "3").replace("@STATS@", "0") % REPO_ROOT
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600, cwd=REPO_ROOT)
    assert r.returncode == 0 and "VARIANT-OK" in r.stdout, (knob, r.stdout[-300:], r.stderr[-1500:])


@pytest.mark.parametrize("overlap,stats", [("0", 0), ("1", 1), ("", 0)])
def test_shipped_library_options_keep_parity(overlap, stats):
    """The shipped library's two options: VQA_OVERLAP (the environment's initial value of VQA_OPT_OVERLAP; default on)
    and VQA_OPT_HYST_STATS.  Neither changes a result."""
    import subprocess
    import sys
    env = dict(os.environ)
    env.pop("VQA_LIB_PATH", None)
    env.pop("VQA_OVERLAP", None)
    if overlap:
        env["VQA_OVERLAP"] = overlap
    code = _VARIANT_CODE.replace("@FLAVOUR@", "0").replace("@STATS@", str(stats)) % REPO_ROOT
    code = code.replace("eng = rtvqa_amd.Engine(0)\n", "eng = rtvqa_amd.Engine(0)\nassert eng.get_option(N.OPT_OVERLAP) == %d\n" % (0 if overlap == "0" else 1))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600, cwd=REPO_ROOT)
    assert r.returncode == 0 and "VARIANT-OK" in r.stdout, (overlap, stats, r.stdout[-300:], r.stderr[-1500:])


def test_overlap_option_toggles_within_one_context(engine):
    """VQA_OPT_OVERLAP flips between submits of one ctx; records are identical (only the diagnostic hyst_steps may differ)."""
    from rtvqa_amd import _native as N
    fr = _frames("natural", 6, 270, 480, seed=31)
    assert engine.get_option(N.OPT_OVERLAP) == 1  # the product's default
    try:
        a = engine.complexity(fr[1:], prev0=fr[0], mask=N.M_ALL, dct_mode=N.DCT_BLOCK8)
        engine.set_overlap(False)
        b = engine.complexity(fr[1:], prev0=fr[0], mask=N.M_ALL, dct_mode=N.DCT_BLOCK8)
        engine.set_overlap(True)
        c = engine.complexity(fr[1:], prev0=fr[0], mask=N.M_ALL, dct_mode=N.DCT_BLOCK8)
        assert _stable(a) == _stable(b) == _stable(c)
        engine.complexity_submit(fr[1:], prev0=fr[0], mask=N.M_EDGE)
        assert engine.lib.vqa_set_option(engine.ctx, N.OPT_OVERLAP, 0) == N.VQA_ERR_STATE  # not while a batch is pending
        engine.complexity_wait()
        assert engine.lib.vqa_set_option(engine.ctx, 99, 0) == N.VQA_ERR_INVALID
    finally:
        engine.set_overlap(True)


_DRAIN_CODE = (
    "import sys, numpy as np; sys.path.insert(0, %r)\n"
    "import rtvqa_amd\n"
    "from rtvqa_amd import _native as N, synth\n"
    "from rtvqa_amd.engine import bgr_planes\n"
    "from oracle import c_oracle as co\n"
    "from oracle import pipeline as pl\n"
    "h, w = 270, 480\n"
    "fr = synth.s_natural(5, h, w, seed=17)\n"
    "eng = rtvqa_amd.Engine(0)\n"
    "assert eng.lib.vqa_build_flavour() & N.FLAVOUR_TEST_SEAMS\n"
    "eng.set_overlap(@OVERLAP@)\n"
    "failed = 0\n"
    "for attempt in range(4):\n"   # the N-th reservation of the ctx fails exactly once; every other call must be exact
    "    try:\n"
    "        rec = eng.complexity(fr[1:], prev0=fr[0], mask=N.M_ALL, dct_mode=N.DCT_BLOCK8)\n"
    "        q = eng.quality(fr[:2], fr[1:3], bgr_planes(h, w), N.SSIM_GAUSS)\n"
    "    except N.VqaError as e:\n"
    "        assert e.status == N.VQA_ERR_OOM and 'test seam' in str(e), str(e)\n"
    "        failed += 1\n"
    "        eng._pending_c = eng._pending_q = None\n"
    "        buf = (N.VqaFrameMetrics * 4)()\n"
    "        assert eng.lib.vqa_complexity_wait(eng.ctx, buf, 4) == N.VQA_ERR_STATE  # nothing is pending after a failed submit\n"
    "        continue\n"
    "    for i in range(4):\n"
    "        g, gp = co.bgr2gray(fr[i + 1]), co.bgr2gray(fr[i])\n"
    "        assert (rec[i]['hist_gray'] == co.hist_u8(g)).all(), ('hist', attempt, i)\n"
    "        assert int(rec[i]['edge_count']) == co.canny(g, 100, 200)[0] and not rec[i]['hyst_overflow'], ('canny', attempt, i)\n"
    "        nb, sad, hist = co.block_sad(gp, g, 7)\n"
    "        assert int(rec[i]['sad_sum']) == sad and (rec[i]['mv_d2_hist'] == hist).all(), ('sad', attempt, i)\n"
    "        e, l1, _ = co.dct8x8(gp, g)\n"
    "        assert abs(rec[i]['dct_energy'] - e) <= 1e-4 * e and abs(rec[i]['temporal_dct_l1'] - l1) <= 1e-4 * l1, ('dct', attempt, i)\n"
    "    sse, ssim = pl.frame_quality(fr[0], fr[1], bgr_planes(h, w), 'gauss')\n"
    "    for p in range(3):\n"
    "        assert int(q[0, p]['sse']) == sse[p] and abs(q[0, p]['ssim'] - ssim[p]) <= 1e-4 * abs(ssim[p]), ('ssim', attempt, p)\n"
    "print('DRAIN-OK', failed)\n"
)


@pytest.mark.parametrize("overlap", [1, 0])
@pytest.mark.parametrize("fail_at", [1, 3, 4, 5, 6, 8, 10, 11, 12, 14, 15, 18, 20, 27, 32])
def test_failed_submit_is_drained_and_the_context_stays_usable(fail_at, overlap):
    """LAB build, VQA_FAIL_ENSURE_AT=N: the N-th device reservation of the context (scratch buffer or table array)
    reports VQA_ERR_OOM.  A full-suite
    complexity submit from host frames makes 14 reservations the first time (1-2 staging, copies already enqueued behind
    them; 3 results; 4 gray planes; 5 DCT partials, gray + histogram kernels enqueued; 6-10 Canny's, AFTER the fork:
    block-SAD is then in flight on a side stream; 11-14 the four arrays of ORB's resize table - a failure at 12 leaves
    a half-built table, which must be freed and rebuilt by the next submit), the quality submit 4 more (15-18); from the
    second round on the table is cached (10 + 4 per round: 19-32 ...) and nothing is reallocated.  The failing submit must
    return the error with
    nothing pending and nothing in flight, and the SAME context must then return oracle-exact records, with the
    side-stream overlap on and off (video_processing.py:295-297: log, re-raise, nothing left running)."""
    import subprocess
    import sys
    from rtvqa_amd import _native as N
    env = dict(os.environ, VQA_LIB_PATH=N.LAB_LIB_PATH, VQA_FAIL_ENSURE_AT=str(fail_at))
    code = _DRAIN_CODE.replace("@OVERLAP@", str(overlap)) % REPO_ROOT
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600, cwd=REPO_ROOT)
    assert r.returncode == 0 and "DRAIN-OK 1" in r.stdout, (fail_at, overlap, r.stdout[-300:], r.stderr[-1500:])


_DRAIN_TABLES_CODE = (
    "import sys, numpy as np; sys.path.insert(0, %r)\n"
    "import rtvqa_amd\n"
    "from rtvqa_amd import _native as N, synth\n"
    "from oracle import c_oracle as co\n"
    "h, w = 270, 480\n"
    "fr = synth.s_natural(3, h, w, seed=23)\n"
    "eng = rtvqa_amd.Engine(0)\n"
    "assert eng.lib.vqa_build_flavour() & N.FLAVOUR_TEST_SEAMS\n"
    "failed = 0\n"
    "for attempt in range(3):\n"
    "    try:\n"   # the full-frame DCT by FFT passes needs two plans of two device arrays each; Farneback three levels of tables
    "        rec = eng.complexity(fr[1:], prev0=fr[0], mask=N.M_DCT | N.M_TEMPORAL_DCT | N.M_MOTION, dct_mode=N.DCT_FULL,\n"
    "                             motion_mode=N.MOTION_FARNEBACK)\n"
    "    except N.VqaError as e:\n"
    "        assert e.status == N.VQA_ERR_OOM and 'test seam' in str(e), str(e)\n"
    "        failed += 1\n"
    "        eng._pending_c = None\n"
    "        continue\n"
    "    for i in range(2):\n"
    "        g, gp = co.bgr2gray(fr[i + 1]), co.bgr2gray(fr[i])\n"
    "        e, l1 = co.dct_energy_full(g), co.temporal_dct_full(gp, g)\n"
    "        assert abs(rec[i]['dct_energy'] - e) <= 1e-4 * e and abs(rec[i]['temporal_dct_l1'] - l1) <= 1e-4 * l1, ('dct', attempt, i)\n"
    "        f = co.farneback(gp, g)\n"
    "        assert abs(rec[i]['flow_mag_mean'] - f) <= 1e-4 * f, ('flow', attempt, i)\n"
    "print('DRAIN-OK', failed)\n"
)


@pytest.mark.parametrize("fail_at", [5, 6, 7, 8, 12, 17, 25])
def test_half_built_tables_are_freed_and_rebuilt(fail_at):
    """LAB build: the reservation that fails is one of a table set's arrays (the FFT plans' twiddle / post-twiddle pairs,
    the Farneback pyramid's resize tables: uploads 5.. of this submit).  The set's earlier arrays must not leak into the
    cache half-built: the failing submit reports the error, the next one rebuilds the set and is oracle-exact."""
    import subprocess
    import sys
    from rtvqa_amd import _native as N
    env = dict(os.environ, VQA_LIB_PATH=N.LAB_LIB_PATH, VQA_FAIL_ENSURE_AT=str(fail_at))
    r = subprocess.run([sys.executable, "-c", _DRAIN_TABLES_CODE % REPO_ROOT], env=env, capture_output=True, text=True,
                       timeout=600, cwd=REPO_ROOT)
    assert r.returncode == 0 and "DRAIN-OK 1" in r.stdout, (fail_at, r.stdout[-300:], r.stderr[-1500:])


_HYST_CODE = (
    "import json, os, sys, numpy as np; sys.path.insert(0, %r)\n"
    "import rtvqa_amd\n"
    "from rtvqa_amd import _native as N\n"
    "from oracle import c_oracle as co\n"
    "r = np.full((80, 4000), 20, np.uint8); r[40, :] = 50; r[40, 5] = 255\n"      # a weak line 62 tiles long, one strong seed
    "rng = np.random.default_rng(5); noise = rng.integers(0, 256, (80, 4000), dtype=np.uint8)\n"
    "fr = np.repeat(np.stack([noise, r, noise[::-1].copy()])[..., None], 3, 3)\n"
    "want = [int(co.canny(g, 100, 200)[0]) for g in (noise, r, noise[::-1].copy())]\n"
    "eng = rtvqa_amd.Engine(0)\n"
    "try:\n"
    "    rec = eng.complexity(fr, mask=N.M_EDGE)\n"
    "    print('RESULT', json.dumps([[int(v) for v in rec['hyst_overflow']], [int(v) for v in rec['edge_count']], want]))\n"
    "except N.VqaError as e:\n"
    "    print('REFUSED', e.status, str(e))\n"
    "    for k in ('VQA_HYST_MAX_ROUNDS', 'VQA_HYST_RESCUE_MAX_ROUNDS'): os.environ.pop(k, None)\n"
    "    rec = eng.complexity(fr[:1], mask=N.M_EDGE)\n"                             # the refusing context stays usable ...
    "    e2 = rtvqa_amd.Engine(0)\n"                                                # ... and a context without the seams finishes the batch
    "    rec2 = e2.complexity(fr, mask=N.M_EDGE)\n"
    "    print('AFTER', int(rec[0]['edge_count']) == want[0], [int(v) for v in rec2['hyst_overflow']], [int(v) for v in rec2['edge_count']] == want)\n"
)


def test_hysteresis_overflow_is_completed_or_refused_never_undercounted():
    """north_star: edge counts are bit-exact.  The tail's round bound exists so the grid always drains; a frame that hits it is
    finished ON THE DEVICE by the rescue pass (hyst_overflow = 2, edge_count exact = the oracle's), and if even that stops
    short vqa_complexity_wait FAILS (VQA_ERR_INCOMPLETE, records zeroed) - a lower bound is never returned as a count.
    Both forced with the LAB build's seams (VQA_HYST_MAX_ROUNDS, VQA_HYST_RESCUE_MAX_ROUNDS; subprocesses); the shipped
    library has no such switches and must report the full count with the flag clear."""
    import subprocess
    import sys
    from rtvqa_amd import _native as N

    def run(lab, **seams):
        env = dict(os.environ)
        for k in ("VQA_HYST_MAX_ROUNDS", "VQA_HYST_RESCUE_MAX_ROUNDS", "VQA_LIB_PATH"):
            env.pop(k, None)
        if lab:
            env["VQA_LIB_PATH"] = N.LAB_LIB_PATH
        env.update(seams)
        r = subprocess.run([sys.executable, "-c", _HYST_CODE % REPO_ROOT], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-800:]
        return [l for l in r.stdout.splitlines() if l.startswith(("RESULT", "REFUSED", "AFTER"))]

    import json
    plain = run(True)
    assert len(plain) == 1 and plain[0].startswith("RESULT "), plain
    flags, got, want = json.loads(plain[0][len("RESULT "):])
    assert flags == [0, 0, 0] and got == want and got[1] > 3900, plain     # (the whole weak line is an edge)
    # the tail stops after ONE round: the 62-tile chain is nowhere near done - the rescue pass finishes it, and says so
    rescued = run(True, VQA_HYST_MAX_ROUNDS="1")
    assert len(rescued) == 1 and rescued[0].startswith("RESULT "), rescued
    flags2, got2, want2 = json.loads(rescued[0][len("RESULT "):])
    assert got2 == want2 == want and flags2[1] == 2 and set(flags2) <= {0, 2}, rescued
    # both passes bounded: no count at all, an error that names the frame, and the context stays usable
    refused = run(True, VQA_HYST_MAX_ROUNDS="1", VQA_HYST_RESCUE_MAX_ROUNDS="1")
    assert len(refused) == 2 and refused[0].startswith("REFUSED %d " % N.VQA_ERR_INCOMPLETE), refused
    assert "frame 1 of 3" in refused[0] and "lower bound" in refused[0], refused
    assert refused[1] == "AFTER True [0, 0, 0] True", refused
    # the shipped library ignores the variables
    shipped = run(False, VQA_HYST_MAX_ROUNDS="1", VQA_HYST_RESCUE_MAX_ROUNDS="1")
    assert shipped == plain, (shipped, plain)


def test_results_are_bit_identical_run_to_run(engine):
    """Every float sum goes through per-block partials and a fixed-order finalize, every integer through exact atomics:
    the same batch must return the same bytes every time - alone, and while a second context keeps the chip busy with
    other kernels (waves of one workgroup then drift apart, which is what exposes a missing barrier).  Round 4's first
    two-rows-per-barrier SSIM kernel had a read/overwrite race on an LDS row buffer that passed every parity test;
    the bench's serial-versus-timed comparison caught it, and this test is its permanent form."""
    import rtvqa_amd
    from rtvqa_amd import _native as N
    from rtvqa_amd.engine import bgr_planes
    h, w, n = 1080, 1920, 24
    d = engine.upload(_frames("natural", n + 1, h, w, seed=41))
    ref, dist = d.slice(0, n), d.slice(1, n + 1)
    planes = bgr_planes(h, w)
    q0 = engine.quality(ref, dist, planes, N.SSIM_GAUSS)
    f0 = engine.quality(ref, dist, planes, N.SSIM_FFMPEG)
    c0 = engine.complexity(dist, prev0=d.frame(0), mask=N.M_ALL, dct_mode=N.DCT_BLOCK8)
    with rtvqa_amd.Engine(engine.device) as other:
        for i in range(12):
            busy = i % 3 != 0
            if busy:  # the other context's full suite runs next to the kernels under test
                other.complexity_submit(dist, d.frame(0), N.M_ALL, other.make_params(dct_mode=N.DCT_BLOCK8))
            q = engine.quality(ref, dist, planes, N.SSIM_GAUSS)
            f = engine.quality(ref, dist, planes, N.SSIM_FFMPEG) if i % 4 == 1 else None
            if busy:
                c = other.complexity_wait()
                assert _stable(c) == _stable(c0), i
                other.quality_submit(ref, dist, planes, N.SSIM_GAUSS)
                c = engine.complexity(dist, prev0=d.frame(0), mask=N.M_ALL, dct_mode=N.DCT_BLOCK8)
                q2 = other.quality_wait()
                assert _stable(c) == _stable(c0), i
                assert (q2["sse"].tobytes(), q2["ssim"].tobytes()) == (q0["sse"].tobytes(), q0["ssim"].tobytes()), i
            assert (q["sse"].tobytes(), q["ssim"].tobytes()) == (q0["sse"].tobytes(), q0["ssim"].tobytes()), i
            if f is not None:
                assert f["ssim"].tobytes() == f0["ssim"].tobytes(), i


def test_farneback_and_full_dct_are_bit_identical_run_to_run_and_across_the_overlap_option(engine):
    """The kernels with the hardest synchronisation - the fused Farneback iteration's one-barrier double-buffered LDS, the
    in-place Stockham passes of the FFT-based full-frame DCT, the Farneback pre-passes piped onto a side stream through
    events (with the chunk-seam wait) and the full-frame DCT forked onto its own stream - must return the same BYTES every
    time: alone, next to a busy second context, and with VQA_OPT_OVERLAP on and off.  (Round 4's SSIM LDS race passed every
    parity test and was caught only by a byte comparison.)  The chunk seam has its own test below."""
    import rtvqa_amd
    from rtvqa_amd import _native as N
    fields = ("flow_mag_mean", "dct_energy", "temporal_dct_l1", "edge_count", "sad_sum")
    params = engine.make_params(dct_mode=N.DCT_FULL, motion_mode=N.MOTION_FARNEBACK)

    def run(d, n):
        return engine.complexity(d.slice(1, n + 1), prev0=d.frame(0), mask=N.M_ALL, params=params)

    def same(a, b):
        return all(a[f].tobytes() == b[f].tobytes() for f in fields)

    keep = engine.get_option(N.OPT_OVERLAP)
    try:
        # 1080p: FFT passes for the DCT (1080 = 2^3 3^3 5, 1920 = 2^7 3 5), one Farneback chunk
        n = 6
        d = engine.upload(_frames("natural", n + 1, 1080, 1920, seed=43))
        engine.set_overlap(True)
        base = run(d, n)
        assert base["flow_mag_mean"].min() > 0 and base["temporal_dct_l1"].min() > 0
        with rtvqa_amd.Engine(engine.device) as other:
            po = other.make_params(dct_mode=N.DCT_BLOCK8)
            for i in range(6):
                engine.set_overlap(i % 2 == 0)
                busy = i >= 2
                if busy:
                    other.complexity_submit(d.slice(1, n + 1), d.frame(0), N.M_ALL, po)
                got = run(d, n)
                if busy:
                    other.complexity_wait()
                assert same(got, base), (i, [f for f in fields if got[f].tobytes() != base[f].tobytes()])
        d._owner.free()
    finally:
        engine.set_overlap(bool(keep))


_FB_SEAM_CODE = (
    "import sys, numpy as np; sys.path.insert(0, %r)\n"
    "import rtvqa_amd\n"
    "from rtvqa_amd import _native as N, synth\n"
    "n, h, w = 23, 240, 426\n"
    "fr = synth.s_noise(n + 1, h, w, seed=44)\n"
    "fields = ('flow_mag_mean', 'dct_energy', 'temporal_dct_l1', 'edge_count')\n"
    "out = []\n"
    "with rtvqa_amd.Engine(0) as eng, rtvqa_amd.Engine(0) as other:\n"
    "    assert eng.lib.vqa_build_flavour() & N.FLAVOUR_TEST_SEAMS\n"
    "    d = eng.upload(fr)\n"
    "    p = eng.make_params(dct_mode=N.DCT_FULL, motion_mode=N.MOTION_FARNEBACK)\n"
    "    for i in range(6):\n"
    "        eng.set_overlap(i %% 2 == 0)\n"
    "        if i >= 2:\n"
    "            other.complexity_submit(d.slice(1, n + 1), d.frame(0), N.M_ALL, other.make_params(dct_mode=N.DCT_BLOCK8))\n"
    "        rec = eng.complexity(d.slice(1, n + 1), prev0=d.frame(0), mask=N.M_ALL, params=p)\n"
    "        if i >= 2:\n"
    "            other.complexity_wait()\n"
    "        out.append(tuple(rec[f].tobytes() for f in fields))\n"
    "assert all(o == out[0] for o in out), [k for k, o in enumerate(out) if o != out[0]]\n"
    "np.save(sys.argv[1], np.frombuffer(out[0][0], np.float64))\n"
    "print('SEAM-OK')\n"
)


def test_farneback_chunk_seam_is_bit_stable(tmp_path):
    """A batch that spans Farneback chunks (LAB build, VQA_FB_CHUNK_BYTES shrinks the 12 GiB chunk budget so that 23 pairs of
    240x426 take 1, 2 and 5 chunks): the side stream's pre-passes of chunk k + 1 wait for the iterations of chunk k through
    an event.  With the overlap option on and off, alone and next to a busy context, one chunking always returns the same
    bytes - and, since round 6, different chunkings return the same bytes too (the flow no longer follows the strip geometry)."""
    import subprocess
    import sys
    from rtvqa_amd import _native as N
    per_pair = 59 * 240 * 426
    got = {}
    for chunks, budget in ((1, 0), (2, per_pair * 12), (5, per_pair * 5)):
        env = dict(os.environ, VQA_LIB_PATH=N.LAB_LIB_PATH)
        if budget:
            env["VQA_FB_CHUNK_BYTES"] = str(budget)
        out = str(tmp_path / ("flow%d.npy" % chunks))
        r = subprocess.run([sys.executable, "-c", _FB_SEAM_CODE % REPO_ROOT, out], env=env, capture_output=True, text=True,
                           timeout=600, cwd=REPO_ROOT)
        assert r.returncode == 0 and "SEAM-OK" in r.stdout, (chunks, r.stdout[-300:], r.stderr[-1500:])
        got[chunks] = np.load(out)
    assert got[1].min() > 0
    for chunks in (2, 5):
        assert np.array_equal(got[chunks], got[1]), chunks


def test_region_of_interest_padded_rows(engine):
    """row_stride > 3w: a window inside larger frames, host and device resident, odd (unaligned) origins.
    The host window ends at the very last byte of its parent array, so any read past a row's 3w bytes
    would leave the allocation."""
    from rtvqa_amd import _native as N
    big = _frames("natural", 4, 120, 200, seed=21)
    dbig = engine.upload(big)
    for (y0, y1, x0, x1) in [(7, 120, 13, 200), (0, 64, 0, 128), (31, 95, 5, 69), (1, 120, 1, 200)]:
        sub = big[:, y0:y1, x0:x1]
        want = engine.complexity(np.ascontiguousarray(sub[1:]), prev0=np.ascontiguousarray(sub[0]), mask=N.M_ALL,
                                 dct_mode=N.DCT_BLOCK8)
        got_h = engine.complexity(sub[1:], prev0=sub[0], mask=N.M_ALL, dct_mode=N.DCT_BLOCK8)
        droi = dbig.roi(y0, y1, x0, x1)
        got_d = engine.complexity(droi.slice(1, 4), prev0=droi.frame(0), mask=N.M_ALL, dct_mode=N.DCT_BLOCK8)
        for got in (got_h, got_d):
            assert _stable(got) == _stable(want), (y0, y1, x0, x1)
        g = co.bgr2gray(np.ascontiguousarray(sub[1]))
        assert (want[0]["hist_gray"] == co.hist_u8(g)).all()
        assert int(want[0]["edge_count"]) == co.canny(g, 100, 200)[0]
        # resized path reads the padded source rows too
        want = engine.complexity(np.ascontiguousarray(sub[1:]), mask=N.M_GRAY_HIST | N.M_COLOR_HIST | N.M_DCT, resize=(40, 24))
        got_d = engine.complexity(droi.slice(1, 4), mask=N.M_GRAY_HIST | N.M_COLOR_HIST | N.M_DCT, resize=(40, 24))
        got_h = engine.complexity(sub[1:], mask=N.M_GRAY_HIST | N.M_COLOR_HIST | N.M_DCT, resize=(40, 24))
        assert _stable(got_d) == _stable(want) and _stable(got_h) == _stable(want)


def test_argument_errors(engine):
    from rtvqa_amd import _native as N
    fr = np.zeros((1, 32, 32, 3), np.uint8)
    with pytest.raises(N.VqaError):
        engine.complexity(fr, mask=0)
    with pytest.raises(N.VqaError):
        engine.complexity(fr, mask=N.M_MOTION, sad_range=9)
    with pytest.raises(N.VqaError):
        engine.quality(fr[..., 0], fr[..., 0], [(8, 8, 0, 32, 1)], N.SSIM_GAUSS)  # smaller than the window
    import ctypes as C
    buf = (N.VqaFrameMetrics * 1)()
    assert engine.lib.vqa_complexity_wait(engine.ctx, buf, 1) == N.VQA_ERR_STATE  # wait without submit
