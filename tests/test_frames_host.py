"""Host-side ingest and CSV glue (no GPU): Y4M round trip, raw yuv420p reader, CSV row writer."""
import csv
import os

import numpy as np
import pytest

from rtvqa_amd import frames, synth
from rtvqa_amd import video_processing as vp


def test_y4m_round_trip_and_raw(tmp_path):
    bgr = synth.s_natural(3, 36, 50, seed=1)
    yuv = frames.bgr_to_yuv420p(bgr)
    assert yuv.shape == (3, frames.frame_bytes_yuv420p(36, 50)) == (3, 36 * 50 + 2 * 18 * 25)
    p = str(tmp_path / "a.y4m")
    frames.write_y4m(p, yuv, 36, 50, fps=(30000, 1001))
    arr, h, w, fps = frames.read_y4m(p)
    assert (arr == yuv).all() and (h, w) == (36, 50) and fps == pytest.approx(29.97, abs=1e-2)
    assert frames.read_y4m(p, max_frames=2)[0].shape[0] == 2
    raw = str(tmp_path / "a.yuv")
    yuv.tofile(raw)
    assert (frames.read_raw_yuv420p(raw, 36, 50) == yuv).all()
    bad = tmp_path / "b.y4m"
    bad.write_bytes(b"RIFF....")
    with pytest.raises(ValueError):
        frames.read_y4m(str(bad))
    # gray ramp: luma of B=G=R=v is ((220 v + 128) >> 8) + 16
    g = np.repeat(np.arange(0, 256, 5, dtype=np.uint8)[None, None, :, None], 3, axis=3).repeat(4, 1)
    y = frames.bgr_to_yuv420p(g)[0, :g.shape[1] * g.shape[2]].reshape(g.shape[1], g.shape[2])
    assert (y[0] == ((220 * np.arange(0, 256, 5) + 128) >> 8) + 16).all()


def test_csv_writer_header_once(tmp_path):
    p = str(tmp_path / "q.csv")
    vp.thread_safe_update_csv({"PSNR": 50.78, "SSIM": 0.994884}, p)
    vp.thread_safe_update_csv({"PSNR": 48.13, "SSIM": 0.9}, p)
    rows = list(csv.reader(open(p)))
    assert rows[0] == ["PSNR", "SSIM"] and len(rows) == 3 and rows[2] == ["48.13", "0.9"]


def test_config_loader_matches_reference_checks(tmp_path):
    import json
    good = {"crf": 23, "vmaf_model_path": None, "resize_width": 64, "resize_height": 64, "frame_interval": 10}
    p = tmp_path / "config.json"
    p.write_text(json.dumps(good))
    assert vp.load_config(str(p)) == good
    for bad, msg in (({**good, "crf": 0}, "CRF"), ({**good, "resize_width": 0}, "Resize"),
                     ({**good, "frame_interval": 0}, "Frame interval"), ({**good, "num_workers": 2.5}, "num_workers")):
        with pytest.raises(ValueError, match=msg):
            vp.validate_config(bad)
    with pytest.raises(FileNotFoundError):
        vp.load_config(str(tmp_path / "missing.json"))
