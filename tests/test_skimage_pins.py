"""Third-party pins: outputs of scikit-image 0.18.3 / SciPy 1.7.1 (tests/golden/skimage_pins.json, made by
oracle/gen_pins_skimage.py under /opt/conda/bin/python3.9 in the build container) against the oracle (CPU half) and
against the HIP path (-m gpu half).  Neither library is this repository's code, so these are the pins DESIGN.md §3
calls "pinned by skimage": Gaussian-windowed SSIM (north_star's definition in place of video_processing.py:276),
MSE / PSNR (video_processing.py:275), the entropy tails (complexity_metrics.py:413-414, :467-473), the FAST-9/16
corner test (:386-387), the orthonormal DCT sums (:363-364, :574-579), the Sobel / L1-magnitude stage of Canny (:503), the
bilinear resize GEOMETRY (:359) and BGR2GRAY's channel order and weights (Pillow 8.4.0, within one level).
NOT pinned by them: OpenCV's BGR2GRAY / INTER_LINEAR rounding, Canny's NMS / sector test / hysteresis, Farneback,
FFmpeg's vf_ssim integers.
"""
import json
import math
import os

import numpy as np
import pytest

from oracle import c_oracle as co
from oracle import gen_pins_skimage as gp
from rtvqa_amd import complexity_metrics as cm

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "skimage_pins.json")))
PAIR_CASE = {c[0]: c for c in gp.PAIR_CASES}
FRAME_CASE = {c[0]: c for c in gp.FRAME_CASES}
SSIM_ORACLE_RTOL = 1e-6   # float64 restatement against float64 scikit-image (the judge measured <= 5.3e-7)
SSIM_GPU_RTOL = 1e-4      # north_star's bar for the fp32 kernel
DCT_RTOL = 1e-5           # oracle: f32 transforms against SciPy's f64
GPU_DCT_RTOL = 1e-4

_pairs, _frames = {}, {}


def pair_of(rec):
    if rec["name"] not in _pairs:
        _pairs[rec["name"]] = gp.make_pair(PAIR_CASE[rec["name"]])
    return _pairs[rec["name"]]


def frames_of(rec):
    if rec["name"] not in _frames:
        _frames[rec["name"]] = gp.make_frames(FRAME_CASE[rec["name"]])
    return _frames[rec["name"]]


def rel(a, b):
    return abs(a - b) / max(abs(b), 1e-300)


def pid(rec):
    return rec["name"]


# ---------------------------------------------------------------------------------------- CPU half
def test_fixture_is_third_party_and_inputs_regenerate_bit_identically():
    assert G["versions"]["skimage"] == "0.18.3" and G["versions"]["scipy"].startswith("1.7")
    assert [p["name"] for p in G["pairs"]] == [c[0] for c in gp.PAIR_CASES]
    for rec in G["pairs"]:
        a, b = pair_of(rec)
        assert (gp.sha(a), gp.sha(b)) == (rec["sha_a"], rec["sha_b"]), rec["name"]
        assert a.shape == (rec["h"], rec["w"])
    for rec in G["frames"]:
        fr = frames_of(rec)
        g0, g1 = gp.gray_for(FRAME_CASE[rec["name"]])
        assert (gp.sha(fr[0]), gp.sha(g0), gp.sha(g1)) == (rec["sha_bgr"], rec["sha_gray0"], rec["sha_gray1"])


@pytest.mark.parametrize("rec", G["pairs"], ids=pid)
def test_oracle_ssim_gauss_equals_skimage(rec):
    a, b = pair_of(rec)
    assert rel(co.ssim_gauss(a, b), rec["ssim"]) <= SSIM_ORACLE_RTOL


@pytest.mark.parametrize("rec", G["pairs"], ids=pid)
def test_oracle_sse_and_psnr_equal_skimage(rec):
    from rtvqa_amd.video_processing import _psnr
    a, b = pair_of(rec)
    sse = co.sse_plane(a, b)
    # skimage: mean((a-b)^2) in float64; the SSE is an integer < 2^53, so the division is the only rounding
    assert sse / (rec["h"] * rec["w"]) == rec["mse"]
    if rec["psnr"] == "inf":
        assert sse == 0 and math.isinf(_psnr(0.0))
    else:
        assert rel(_psnr(sse / (rec["h"] * rec["w"])), rec["psnr"]) <= 1e-12


@pytest.mark.parametrize("rec", G["pairs"], ids=pid)
def test_product_gray_entropy_tail_equals_skimage_shannon_entropy(rec):
    """complexity_metrics.py:413-414 in float32 against scipy.stats.entropy(base=2) in float64."""
    for plane, want in zip(pair_of(rec), (rec["entropy_a"], rec["entropy_b"])):
        got = float(cm._gray_entropy(co.hist_u8(plane)))
        assert abs(got - want) <= 1e-6 * max(want, 1.0)


@pytest.mark.parametrize("rec", G["frames"], ids=pid)
def test_product_entropy_tails_on_frames_equal_skimage(rec):
    fr = frames_of(rec)[0]
    g0, _ = gp.gray_for(FRAME_CASE[rec["name"]])
    assert abs(float(cm._gray_entropy(co.hist_u8(g0))) - rec["gray_entropy"]) <= 1e-6 * rec["gray_entropy"]
    counts = np.stack([co.hist_u8(np.ascontiguousarray(fr[..., c])) for c in range(3)])
    # :471-473 adds 1e-8 inside the log2 (<= 768 * 1.44e-8 below the plain entropy sum) and sums in float32
    assert abs(float(cm._color_entropy(counts)) - rec["color_entropy_sum"]) <= 2e-5


@pytest.mark.parametrize("rec", G["frames"], ids=pid)
def test_oracle_fast9_detection_equals_skimage_corner_fast(rec):
    """The segment test of cv2's FAST (9 contiguous of 16, |diff| > 20, 3-pixel border skipped), NMS off."""
    g0, _ = gp.gray_for(FRAME_CASE[rec["name"]])
    n, _score, keep = co.fast9(g0, 20, False)
    ys, xs = np.nonzero(keep)
    assert n == rec["fast9_count"] == len(ys)
    assert int(np.sum((ys.astype(np.int64) * 7919 + xs.astype(np.int64) * 104729) % 1000003)) == rec["fast9_crc"]


@pytest.mark.parametrize("rec", G["frames"], ids=pid)
def test_oracle_bgr2gray_is_bt601_luma_of_the_right_channels(rec):
    """cv2.cvtColor(BGR2GRAY): within one grey level of Pillow's convert("L") and of 0.299 R + 0.587 G + 0.114 B on the same
    pixels (values recorded by the conda interpreter for the plane whose SHA-256 the fixture holds).  Swapped channels or
    other weights are off by tens of levels on these frames; OpenCV's 15-bit rounding itself is not pinned."""
    fr = frames_of(rec)[0]
    g0, _ = gp.gray_for(FRAME_CASE[rec["name"]])
    assert rec["gray_vs_pillow_maxdiff"] <= 1 and rec["gray_vs_float_maxdiff"] < 1.0
    # what the fixture compared IS this oracle's plane (same SHA-256), and the float formula evaluated here agrees with it
    assert gp.sha(g0) == rec["sha_gray0"]
    lum_f = 0.299 * fr[..., 2].astype(np.float64) + 0.587 * fr[..., 1] + 0.114 * fr[..., 0]
    assert float(lum_f.sum()) == pytest.approx(rec["float_luma_sum"], rel=1e-12)
    assert np.abs(lum_f - g0).max() < 1.0
    swapped = co.bgr2gray(np.ascontiguousarray(fr[..., ::-1]))  # R and B exchanged: must NOT pass the same bar
    if rec["generator"] == "natural":
        assert np.abs(lum_f - swapped).max() > 3.0


@pytest.mark.gpu
@pytest.mark.parametrize("rec", G["frames"], ids=pid)
def test_gpu_gray_plane_is_the_pinned_plane(engine, rec):
    """The device's gray plane (vqa_debug_read_plane) is byte-identical to the plane the Pillow / float-luma pins were taken on."""
    from rtvqa_amd import _native as N
    fr = frames_of(rec)
    engine.complexity(fr[:1], mask=N.M_GRAY_HIST)
    h, w = fr.shape[1:3]
    assert gp.sha(engine.debug_plane(3, 0, h, w)) == rec["sha_gray0"]


@pytest.mark.parametrize("rec", G["frames"], ids=pid)
def test_oracle_sobel_stage_of_canny_equals_scipy_ndimage(rec):
    """cv2.Canny's gradient: 3x3 Sobel on a replicated border, |dx| + |dy| - every pixel, borders included, against
    scipy.ndimage.sobel(mode="nearest").  (NMS, the TG22 sector test and the hysteresis stay unpinned.)"""
    g0, _ = gp.gray_for(FRAME_CASE[rec["name"]])
    dx, dy, mag = co.sobel_l1(g0)
    assert (np.abs(dx.astype(np.int32)) + np.abs(dy.astype(np.int32)) == mag).all()
    h, w = mag.shape
    wts = (np.arange(h, dtype=np.int64)[:, None] * 31 + np.arange(w, dtype=np.int64)[None, :] * 17 + 1) % 1009
    assert int(mag.sum()) == rec["sobel_l1_sum"] and int(mag.max()) == rec["sobel_l1_max"]
    assert int((mag.astype(np.int64) * wts).sum()) == rec["sobel_l1_crc"]
    assert int(mag[0].sum() + mag[-1].sum() + mag[:, 0].sum() + mag[:, -1].sum()) == rec["sobel_l1_border_sum"]


@pytest.mark.parametrize("rec", G["frames"], ids=pid)
def test_oracle_dct_sums_equal_scipy(rec):
    g0, g1 = gp.gray_for(FRAME_CASE[rec["name"]])
    assert rel(co.dct_energy_full(g1), rec["dct_full_energy"]) <= DCT_RTOL
    assert rel(co.temporal_dct_full(g0, g1), rec["dct_full_l1"]) <= DCT_RTOL
    e8, l8, _ = co.dct8x8(g0, g1)
    assert rel(e8, rec["dct8_energy"]) <= DCT_RTOL and rel(l8, rec["dct8_l1"]) <= DCT_RTOL
    assert rec["dct8_energy"] == pytest.approx(float((g1.astype(np.float64) ** 2).sum()), rel=1e-12)  # Parseval


@pytest.mark.parametrize("rec", G["frames"], ids=pid)
def test_oracle_resize_geometry_within_one_level_of_float_bilinear(rec):
    """cv2.resize INTER_LINEAR samples at (dx + 0.5) * scale - 0.5 with edge clamping; its 11-bit weights and two
    roundings keep it within one grey level of the float64 interpolation skimage computes at the same positions.
    A wrong centre convention or clamp shows up as errors of many levels on these textures."""
    g0, _ = gp.gray_for(FRAME_CASE[rec["name"]])
    dw, dh = rec["resize_to"]
    want = np.array(rec["resize_float"]).reshape(dh, dw)
    got = co.resize_linear(g0, dw, dh).astype(np.float64)
    assert np.abs(got - want).max() < 1.0
    assert np.abs(got - want).mean() < 0.3


def _area2_fields(plane):
    he, we = plane.shape
    w2 = (np.arange(he, dtype=np.int64)[:, None] * 31 + np.arange(we, dtype=np.int64)[None, :] * 17 + 1) % 1009
    p = plane.astype(np.int64)
    return int(p.sum()), int(p.max()), int((p * w2).sum())


@pytest.mark.parametrize("rec", G["frames"], ids=pid)
def test_oracle_exact_halving_equals_rounded_local_means(rec):
    """cv2.resize(INTER_LINEAR) at an exact factor of two takes INTER_AREA's shortcut: the 2x2 mean rounded half up,
    (a + b + c + d + 2) >> 2.  scikit-image's downscale_local_mean gives the float means; floor(mean + 0.5) must be the
    oracle's plane EXACTLY (sum, maximum and a position-weighted checksum of every pixel)."""
    g0, _ = gp.gray_for(FRAME_CASE[rec["name"]])
    he, we = g0.shape[0] & ~1, g0.shape[1] & ~1
    got = co.resize_linear(np.ascontiguousarray(g0[:he, :we]), we // 2, he // 2)
    assert _area2_fields(got) == (rec["area2_sum"], rec["area2_max"], rec["area2_crc"])


# ---------------------------------------------------------------------------------------- GPU half
@pytest.mark.gpu
@pytest.mark.parametrize("rec", G["pairs"], ids=pid)
def test_gpu_quality_kernel_equals_skimage(engine, rec):
    """k_ssim_gauss (60 % of the c3 step) against scikit-image: SSE exact, SSIM within north_star's 1e-4."""
    from rtvqa_amd import _native as N
    from rtvqa_amd.engine import gray_planes
    a, b = pair_of(rec)
    h, w = a.shape
    q = engine.quality(a[None], b[None], gray_planes(h, w), N.SSIM_GAUSS)
    assert int(q[0, 0]["sse"]) / (h * w) == rec["mse"]
    assert rel(float(q[0, 0]["ssim"]), rec["ssim"]) <= SSIM_GPU_RTOL


@pytest.mark.gpu
def test_gpu_quality_kernel_equals_skimage_in_one_ragged_batch(engine):
    """The same pins through the batched entry: the three 72x104 / 97x131 / 270x480 groups as 2-frame batches."""
    from rtvqa_amd import _native as N
    from rtvqa_amd.engine import gray_planes
    by_geom = {}
    for rec in G["pairs"]:
        by_geom.setdefault((rec["h"], rec["w"]), []).append(rec)
    for (h, w), recs in by_geom.items():
        a = np.stack([pair_of(r)[0] for r in recs])
        b = np.stack([pair_of(r)[1] for r in recs])
        q = engine.quality(a, b, gray_planes(h, w), N.SSIM_GAUSS)
        for i, r in enumerate(recs):
            assert int(q[i, 0]["sse"]) / (h * w) == r["mse"], r["name"]
            assert rel(float(q[i, 0]["ssim"]), r["ssim"]) <= SSIM_GPU_RTOL, r["name"]


@pytest.mark.gpu
@pytest.mark.parametrize("rec", G["frames"], ids=pid)
def test_gpu_histogram_and_dct_kernels_equal_skimage_and_scipy(engine, rec):
    from rtvqa_amd import _native as N
    fr = frames_of(rec)
    mask = N.M_GRAY_HIST | N.M_COLOR_HIST | N.M_DCT | N.M_TEMPORAL_DCT
    r8 = engine.complexity(fr[1:], prev0=fr[0], mask=mask, dct_mode=N.DCT_BLOCK8)[0]
    rf = engine.complexity(fr[1:], prev0=fr[0], mask=mask, dct_mode=N.DCT_FULL)[0]
    assert rel(float(r8["dct_energy"]), rec["dct8_energy"]) <= GPU_DCT_RTOL
    assert rel(float(r8["temporal_dct_l1"]), rec["dct8_l1"]) <= GPU_DCT_RTOL
    assert rel(float(rf["dct_energy"]), rec["dct_full_energy"]) <= GPU_DCT_RTOL
    assert rel(float(rf["temporal_dct_l1"]), rec["dct_full_l1"]) <= GPU_DCT_RTOL
    r0 = engine.complexity(fr[:1], mask=N.M_GRAY_HIST | N.M_COLOR_HIST)[0]
    assert abs(float(cm._gray_entropy(r0["hist_gray"])) - rec["gray_entropy"]) <= 1e-6 * rec["gray_entropy"]
    assert abs(float(cm._color_entropy(r0["hist_bgr"])) - rec["color_entropy_sum"]) <= 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("rec", G["frames"], ids=pid)
def test_gpu_exact_halving_equals_rounded_local_means(engine, rec):
    """The device's resize at an exact factor of two (plane A = resize(gray(frame)), vqa_debug_read_plane) against the same
    scikit-image pin: the rounded 2x2 means, every pixel."""
    from rtvqa_amd import _native as N
    fr = frames_of(rec)
    he, we = fr.shape[1] & ~1, fr.shape[2] & ~1
    crop = np.ascontiguousarray(fr[:1, :he, :we])
    engine.complexity(crop, mask=N.M_DCT, resize=(we // 2, he // 2), dct_mode=N.DCT_BLOCK8)
    got = engine.debug_plane(0, 0, he // 2, we // 2)
    assert _area2_fields(got) == (rec["area2_sum"], rec["area2_max"], rec["area2_crc"])
