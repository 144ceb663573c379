"""Host-side math of the product (and the oracle's EWM) against golden vectors produced by the
REAL reference's cv2-free functions (tests/golden/host_math.json, made by oracle/gen_golden.py)."""
import json
import math
import os

import numpy as np
import pytest

from oracle import np_oracle as no
from rtvqa_amd import complexity_metrics as cm
from rtvqa_amd import pooling
from rtvqa_amd import video_processing as vp

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "host_math.json")))


@pytest.mark.parametrize("case", G["smooth_data"], ids=lambda c: "n%d_a%s" % (len(c["data"]), c["alpha"]))
def test_smooth_data_matches_pandas(case):
    for impl in (cm.smooth_data, pooling.smooth_data, no.ewm_mean):
        sm = impl(case["data"], case["alpha"])
        assert len(sm) == len(case["smoothed"])
        assert np.allclose(sm, case["smoothed"], rtol=1e-13, atol=0)
    if case["mean"] is None:
        assert math.isnan(pooling.pooled_mean(case["data"], case["alpha"]))
    else:
        assert pooling.pooled_mean(case["data"], case["alpha"]) == pytest.approx(case["mean"], rel=1e-13)
        # the linear-functional form used for multi-GPU pooling gives the same number
        c = pooling.pooling_weights(len(case["data"]), case["alpha"])
        assert float(np.dot(c, case["data"])) == pytest.approx(case["mean"], rel=1e-12)


@pytest.mark.parametrize("case", G["process_in_batches_abs"])
def test_process_in_batches_plain_callable(case):
    out = cm.process_in_batches(case["items"], abs, case["num_workers"], batch_size=case["batch_size"])
    assert out == case["out"]


def test_frame_interval_and_normalize():
    for c in G["process_frame_interval"]:
        assert cm.process_frame_interval_for_parallel(tuple(c["timestamps"])) == c["out"]
    for c in G["normalize"]:
        assert cm.normalize(*c["args"]) == c["out"]


def test_scene_complexity_score_table(monkeypatch):
    for c in G["scene_complexity_score"]:
        monkeypatch.setattr(cm, "calculate_average_scene_complexity", lambda *a, _t=tuple(c["metrics_tuple"]), **k: _t)
        assert cm.calculate_scene_complexity_score("x.npy", 64, 64) == pytest.approx(c["score"], rel=1e-15)


def test_extract_metrics_from_logs(tmp_path):
    for k, c in enumerate(G["extract_metrics_from_logs"]):
        p, s = tmp_path / ("p%d.log" % k), tmp_path / ("s%d.log" % k)
        p.write_text(c["psnr_text"])
        s.write_text(c["ssim_text"])
        m = vp.extract_metrics_from_logs(str(p), str(s), str(tmp_path / "absent.json"), "in.mp4", 23, 4486,
                                         "1920x1080", 30.0)
        assert m == c["metrics"]


def test_stats_lines_parse_with_reference_regexes():
    """Lines our writer emits are parsed by the reference's regexes (video_processing.py:160,:166)."""
    import re
    sizes = [(64, 48)] * 3
    line = vp.psnr_stats_line(1, [3072, 6144, 1536], sizes, "rgb")
    assert line.startswith("n:1 mse_avg:1.17 mse_r:1.00 mse_g:2.00 mse_b:0.50 psnr_avg:")
    assert float(re.search(r"psnr_avg:(\s*\d+\.\d+)", line).group(1)) == pytest.approx(10 * math.log10(255 ** 2 / (3.5 / 3)), abs=5e-3)
    # identical frames: 'inf' does not match, key silently absent
    assert re.search(r"psnr_avg:(\s*\d+\.\d+)", vp.psnr_stats_line(1, [0, 0, 0], sizes, "rgb")) is None
    s = vp.ssim_stats_line(1, [0.9, 0.8, 0.7], sizes, "rgb")
    assert s == "n:1 R:0.900000 G:0.800000 B:0.700000 All:0.800000 (6.989700)\n"
    # yuv420p: All is plane-area weighted (4:1:1)
    s = vp.ssim_stats_line(2, [0.9, 0.6, 0.3], [(64, 48), (32, 24), (32, 24)], "yuv")
    assert float(re.search(r"All:(\s*\d+\.\d+)", s).group(1)) == pytest.approx((0.9 * 4 + 0.6 + 0.3) / 6, abs=1e-6)


def test_frame_selection_matches_reference_phases():
    # read_frame_pairs keeps 1-based count % k == 0 (:103-104); timestamps 0-based % k == 0 (:65)
    assert list(cm.selected_indices(45, 10)) == [9, 19, 29, 39]
    assert list(cm.selected_indices(9, 10)) == []
    frames = np.zeros((45, 4, 4, 3), np.uint8)
    frames[:, 0, 0, 0] = np.arange(45)
    pairs = cm.read_frame_pairs(frames, 10)
    assert [(int(c[0, 0, 0]), int(p[0, 0, 0])) for c, p in pairs] == [(19, 9), (29, 19), (39, 29)]
    ts = cm.extract_frame_timestamps(frames, 10, fps=30.0)
    assert len(ts) == 5 and ts[1] == pytest.approx(10000.0 / 30)


def test_bad_sources_raise_like_reference():
    with pytest.raises(ValueError):
        cm.validate_video_path(123)
    with pytest.raises(ValueError):
        cm.validate_video_path("clip.txt")
    from rtvqa_amd import _native as N
    with pytest.raises(N.VqaError):  # no CPU fallback: a kernel call without a device fails loudly
        cm.process_orb_frame_for_parallel(np.zeros((8, 8, 3), np.uint8))


def test_empty_and_too_short_clips_need_no_device():
    """Unopenable video -> [] -> np.mean([]) = NaN in the reference (:56-58, :95-97, :302-309); temporal 0.0 (:541).
    These paths never reach a kernel, so they behave the same on a machine without a GPU."""
    for clip in (np.zeros((0, 8, 8, 3), np.uint8), np.zeros((15, 8, 8, 3), np.uint8), "/nonexistent/clip.npy"):
        out = cm.calculate_average_scene_complexity(clip, 64, 64, frame_interval=10)
        assert len(out) == 8
        assert all(math.isnan(out[i]) for i in (0, 1, 2, 3, 4, 5)) and out[6] == 0.0
    assert cm.read_frame_pairs(np.zeros((15, 8, 8, 3), np.uint8), 10) == []
