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


def test_open_y4m_maps_the_file_and_equals_the_eager_reader(tmp_path):
    """open_y4m: the clip as a strided view of a memory map (frames are a 6-byte FRAME line + bytes apart), byte-identical to what read_y4m
    loads; a stream whose frame headers carry parameters (unequal spacing) falls back to the frame-by-frame reader."""
    bgr = synth.s_natural(5, 36, 50, seed=2)
    yuv = frames.bgr_to_yuv420p(bgr)
    p = str(tmp_path / "m.y4m")
    frames.write_y4m(p, yuv, 36, 50)
    arr, h, w, fps = frames.open_y4m(p)
    fb = frames.frame_bytes_yuv420p(36, 50)
    assert (h, w, fps) == (36, 50, 30.0) and arr.shape == (5, fb) and arr.strides == (fb + 6, 1)
    b = arr
    while b is not None and not isinstance(b, np.memmap):
        b = getattr(b, "base", None)
    assert isinstance(b, np.memmap)          # a view of the mapped file, not a copy
    assert (arr == yuv).all() and (arr == frames.read_y4m(p)[0]).all() and not arr.flags.writeable
    assert frames.open_y4m(p, max_frames=2)[0].shape == (2, fb)
    # a truncated last frame is not a frame
    with open(p, "ab") as f:
        f.write(b"FRAME\n" + bytes(10))
    assert frames.open_y4m(p)[0].shape == (5, fb)
    # frame headers with parameters: not equally spaced -> the parsing reader
    q = str(tmp_path / "params.y4m")
    with open(q, "wb") as f:
        f.write(b"YUV4MPEG2 W50 H36 F25:1 Ip A1:1 C420jpeg\n")
        for k in range(3):
            f.write(b"FRAME Ip\n" if k == 1 else b"FRAME\n")
            f.write(yuv[k].tobytes())
    arr2, _, _, fps2 = frames.open_y4m(q)
    assert arr2.shape == (3, fb) and (arr2 == yuv[:3]).all() and fps2 == 25.0
    empty = str(tmp_path / "empty.y4m")
    with open(empty, "wb") as f:
        f.write(b"YUV4MPEG2 W50 H36 F25:1 C420\n")
    assert frames.open_y4m(empty)[0].shape == (0, fb)
    with pytest.raises(ValueError):
        frames.open_y4m(str(tmp_path / "b.y4m") if (tmp_path / "b.y4m").exists() else _not_y4m(tmp_path))


def _not_y4m(tmp_path):
    p = tmp_path / "not.y4m"
    p.write_bytes(b"RIFF....")
    return str(p)


def test_raw_bgr24_and_raw_yuv_are_mapped(tmp_path):
    """headerless streams (ffmpeg -f rawvideo): packed BGR24 [N,H,W,3] and planar yuv420p, mapped with the geometry the caller
    names; the complexity surface accepts the .bgr24 path once it knows the geometry."""
    from rtvqa_amd import complexity_metrics as cm
    bgr = synth.s_natural(4, 36, 50, seed=3)
    p = str(tmp_path / "clip.bgr24")
    bgr.tofile(p)
    m = frames.open_raw_bgr24(p, 36, 50)
    assert isinstance(m, np.memmap) and m.shape == (4, 36, 50, 3) and (m == bgr).all()
    assert frames.open_raw_bgr24(p, 36, 50, max_frames=3).shape[0] == 3
    with open(p, "ab") as f:
        f.write(bytes(100))                                  # a truncated frame at the end is not a frame
    assert frames.open_raw_bgr24(p, 36, 50).shape[0] == 4
    assert (cm._open_frames(p, 36, 50) == bgr).all() and cm.validate_video_path(p) == "video"
    with pytest.raises(ValueError, match="height and width"):
        cm._open_frames(p)
    yuv = frames.bgr_to_yuv420p(bgr)
    q = str(tmp_path / "clip.yuv")
    yuv.tofile(q)
    arr, layout, h, w = vp._open_quality_stream(q, "bgr24", 36, 50)
    assert layout == "yuv420p" and (h, w) == (36, 50) and isinstance(arr, np.memmap) and (arr == yuv).all()
    with pytest.raises(ValueError, match="height and width"):
        vp._open_quality_stream(q, "yuv420p", None, None)
    good = {"crf": 23, "resize_width": 64, "resize_height": 64, "frame_interval": 10, "height": 36, "width": 50}
    vp.validate_config(good)
    with pytest.raises(ValueError, match="height and width must be positive integers"):
        vp.validate_config(dict(good, height=0))
