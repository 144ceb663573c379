"""process_video_and_extract_metrics - the reference's caller of the hot path (video_processing.py:180-267) - with the
reference's OWN quantities: the quality filters compare decoded yuv420p planes (:274-276, .y4m here), the complexity suite
reads the encoded stream's BGR frames (:242-247, complexity_metrics.py:100), in ONE pass; and every config key this build
adds (ssim_mode, pixfmt, dct_mode, motion, device) against the oracle's reference-shaped pipeline."""
import csv
import re
import threading

import numpy as np
import pytest

from oracle import pipeline as pl

pytestmark = pytest.mark.gpu
RTOL = 1e-4
FIXED = ("Advanced Motion Complexity", "DCT Complexity", "Histogram Complexity", "Edge Detection Complexity",
         "ORB Feature Complexity", "Color Histogram Complexity", "Temporal DCT Complexity", "Framerate Variation")


def _clip(n, h, w, seed=0):
    from rtvqa_amd import synth
    return synth.s_natural(n, h, w, seed=seed)


def _close(a, b, tol=RTOL):
    a, b = float(a), float(b)
    return (a != a and b != b) or abs(a - b) <= tol * max(abs(b), 1e-12)


def _quality_text(ref_frame, dist_frame, planes, layout, ssim_mode):
    """(PSNR, SSIM) as the reference's regexes (video_processing.py:160, :166) read them from the stats lines of frame 1,
    the lines formatted from the ORACLE's numbers"""
    from rtvqa_amd import video_processing as vp
    sse, ssim = pl.frame_quality(ref_frame, dist_frame, planes, ssim_mode)
    sizes = [(p[0], p[1]) for p in planes]
    comps = vp.LAYOUTS[layout][1]
    order = [2, 1, 0] if layout == "bgr24" else list(range(len(comps)))
    names = "rgb" if layout == "bgr24" else comps
    pline = vp.psnr_stats_line(1, [sse[j] for j in order], [sizes[j] for j in order], names)
    sline = vp.ssim_stats_line(1, [ssim[j] for j in order], [sizes[j] for j in order], names)
    mp, ms = re.search(r"psnr_avg:(\s*\d+\.\d+)", pline), re.search(r"All:(\s*\d+\.\d+)", sline)
    return (float(mp.group(1)) if mp else None), (float(ms.group(1)) if ms else None)


def _check_row(m, want, exact=(2, 3, 4, 5, 7)):
    for k, name in enumerate(FIXED):
        assert _close(m[name], want[k], 1e-12 if k in exact else RTOL), (name, m[name], want[k])


@pytest.fixture(scope="module")
def y4m_case(tmp_path_factory):
    from rtvqa_amd import frames, synth
    tmp = tmp_path_factory.mktemp("y4m")
    h, w, n = 120, 168, 26
    ref = _clip(n, h, w, seed=21)
    enc = synth.distort(ref)
    yr, yd = frames.bgr_to_yuv420p(ref), frames.bgr_to_yuv420p(enc)
    pr, pd_, pe = str(tmp / "r.y4m"), str(tmp / "d.y4m"), str(tmp / "enc.npy")
    frames.write_y4m(pr, yr, h, w)
    frames.write_y4m(pd_, yd, h, w)
    np.save(pe, enc)
    return dict(h=h, w=w, n=n, ref=ref, enc=enc, yr=yr, yd=yd, pr=pr, pd=pd_, pe=pe, tmp=tmp)


def test_y4m_pair_with_encoded_bgr_is_the_reference_true_row(y4m_case):
    """(r.y4m, d.y4m, enc.npy) with vf_ssim + Farneback + the reference's full-frame DCT at 64x64 = what the reference itself
    computes for an H.264 clip: psnr_avg text from exact SSE, All: from vf_ssim on the Y, U, V planes, the 8-tuple against the
    oracle pipeline with the same definitions."""
    from rtvqa_amd import video_processing as vp
    from rtvqa_amd.engine import yuv420p_planes
    c = y4m_case
    cfg = {"crf": 23, "resize_width": 64, "resize_height": 64, "frame_interval": 4, "batch_size": 7,
           "ssim_mode": "ffmpeg", "motion": "farneback"}
    out = str(c["tmp"] / "a.csv")
    m = vp.process_video_and_extract_metrics(c["pr"], c["pd"], cfg, csv_file=out, column_order="fixed", encoded_bgr=c["pe"])
    want = pl.calculate_average_scene_complexity(list(c["enc"]), 64, 64, frame_interval=4, dct_mode="full", motion="farneback")
    _check_row(m, want)
    psnr, ssim = _quality_text(c["yr"][0], c["yd"][0], yuv420p_planes(c["h"], c["w"]), "yuv420p", "ffmpeg")
    assert m["PSNR"] == psnr                         # 2-decimal text of an exact integer SSE
    assert abs(m["SSIM"] - ssim) <= 1.5e-6           # 6-decimal text of float sums
    assert m["Resolution (px)"] == "%dx%d" % (c["w"], c["h"]) and m["CRF"] == 23
    rows = list(csv.reader(open(out)))
    assert len(rows) == 2 and rows[0][0] == "Bitrate (kbps)" and "Temporal DCT Complexity" in rows[0]
    # the same call with the streams as arrays (pixfmt names the layout), chunked differently: the same row
    cfg2 = dict(cfg, pixfmt="yuv420p", batch_size=100)
    m2 = vp.process_video_and_extract_metrics(c["yr"], c["yd"], cfg2, csv_file=out, column_order="fixed", encoded_bgr=c["enc"])
    for k in FIXED + ("PSNR", "SSIM"):   # (neither the SSIM mean nor - since round 6 - Farneback's follows the batch geometry)
        assert m2[k] == m[k] or (m2[k] != m2[k] and m[k] != m[k]), k


def test_split_pass_equals_the_two_halves_run_alone_from_every_residence(y4m_case):
    """The one pass over three streams gives, bit for bit, what a quality-only pass over the planar pair and a complexity-only
    pass over the BGR stream give - from pageable, pinned and device memory, and with the streams in DIFFERENT places."""
    from rtvqa_amd import _native as N
    from rtvqa_amd import complexity_metrics as cm
    from rtvqa_amd import stream
    from rtvqa_amd.engine import DeviceFrames, yuv420p_planes
    c = y4m_case
    h, w = c["h"], c["w"]
    eng = cm.get_engine()
    planes = yuv420p_planes(h, w)
    q0, _ = stream.run(c["yd"], c["yr"], quality=stream.Quality(planes, N.SSIM_FFMPEG), batch_size=6)
    _, s0 = stream.run(c["enc"], complexity=stream.Complexity((64, 48), 3), batch_size=6)

    def pinned(a):
        p = eng.alloc_pinned(a.shape)
        p[...] = a
        return p

    def resident(a, planar):
        d = eng.upload(a.reshape(a.shape[0], 1, -1) if planar else a)
        if planar:
            return DeviceFrames(d.ptr, a.shape[0], h, w, frame_stride=a.shape[1], row_stride=w, owner=d, channels=1)
        return d

    homes = {"pageable": lambda a, planar: a, "pinned": lambda a, planar: pinned(a), "device": resident}
    pins = []
    for names in (("pageable",) * 3, ("pinned",) * 3, ("device",) * 3, ("pageable", "device", "pinned"), ("device", "pinned", "pageable")):
        enc, yr, yd = homes[names[0]](c["enc"], False), homes[names[1]](c["yr"], True), homes[names[2]](c["yd"], True)
        pins += [a for a in (enc, yr, yd) if isinstance(a, np.ndarray) and a is not c["enc"] and a is not c["yr"] and a is not c["yd"]]
        if (names[1] == "device") != (names[2] == "device"):
            continue
        for bs in (6, 100):
            q, s = stream.run(enc, yr, quality=stream.Quality(planes, N.SSIM_FFMPEG), complexity=stream.Complexity((64, 48), 3),
                              batch_size=bs, qdist=yd)
            assert (q[0] == q0[0]).all(), names
            assert np.allclose(q[1], q0[1], rtol=0, atol=1e-9), names
            if bs == 6:
                assert (q[1] == q0[1]).all(), names
            for k in stream.KINDS + ("temporal",):
                assert s[k] == s0[k], (names, k)
    for p in pins:
        eng.free_pinned(p)


@pytest.mark.parametrize("key,a,b", [("ssim_mode", "gauss", "ffmpeg"), ("dct_mode", "block8", "full"), ("motion", "sad", "farneback")])
def test_config_key_selects_the_definition(tmp_path, key, a, b):
    """One config key at a time, both of its values, each against the oracle pipeline with the same definition."""
    from rtvqa_amd import synth
    from rtvqa_amd import video_processing as vp
    from rtvqa_amd.engine import bgr_planes
    h, w = 120, 168      # (above 128x128 pixels: dct_mode "auto" means 8x8 blocks here)
    ref = _clip(14, h, w, seed=33)
    enc = synth.distort(ref)
    base = {"crf": 30, "resize_width": w, "resize_height": h, "frame_interval": 2, "batch_size": 5}
    rows = {}
    for val in (a, b):
        cfg = dict(base, **{key: val})
        m = rows[val] = vp.process_video_and_extract_metrics(ref, enc, cfg, csv_file=str(tmp_path / "k.csv"), column_order="fixed")
        dct = {"block8": "block8", "full": "full"}.get(cfg.get("dct_mode"), "block8")   # auto above 128x128 / at native size: 8x8
        want = pl.calculate_average_scene_complexity(list(enc), w, h, frame_interval=2, dct_mode=dct, motion=cfg.get("motion", "sad"))
        _check_row(m, want, exact=(2, 3, 4, 5, 7) + ((0,) if cfg.get("motion", "sad") == "sad" else ()))
        psnr, ssim = _quality_text(ref[0], enc[0], bgr_planes(h, w), "bgr24", cfg.get("ssim_mode", "gauss"))
        assert m["PSNR"] == psnr and abs(m["SSIM"] - ssim) <= (2e-6 if cfg.get("ssim_mode") == "ffmpeg" else 1e-4), (val, m["SSIM"], ssim)
        assert m["CRF"] == 30
    changed = {"ssim_mode": "SSIM", "dct_mode": "DCT Complexity", "motion": "Advanced Motion Complexity"}[key]
    assert rows[a][changed] != rows[b][changed]
    others = [k for k in FIXED + ("PSNR",) if k != changed and not (key == "dct_mode" and k == "Temporal DCT Complexity")]
    assert all(rows[a][k] == rows[b][k] for k in others if not (isinstance(rows[a][k], float) and rows[a][k] != rows[a][k]))


def test_config_key_dct_mode_auto_is_full_frame_at_thumbnail_size(tmp_path):
    from rtvqa_amd import synth
    from rtvqa_amd import video_processing as vp
    ref = _clip(12, 96, 128, seed=34)
    enc = synth.distort(ref)
    cfg = {"crf": 23, "resize_width": 64, "resize_height": 64, "frame_interval": 3}
    auto = vp.process_video_and_extract_metrics(ref, enc, cfg, csv_file=str(tmp_path / "a.csv"), column_order="fixed")
    full = vp.process_video_and_extract_metrics(ref, enc, dict(cfg, dct_mode="full"), csv_file=str(tmp_path / "a.csv"), column_order="fixed")
    blk = vp.process_video_and_extract_metrics(ref, enc, dict(cfg, dct_mode="block8"), csv_file=str(tmp_path / "a.csv"), column_order="fixed")
    assert auto["Temporal DCT Complexity"] == full["Temporal DCT Complexity"] != blk["Temporal DCT Complexity"]
    want = pl.calculate_average_scene_complexity(list(enc), 64, 64, frame_interval=3, dct_mode="block8")
    assert _close(blk["Temporal DCT Complexity"], want[6]) and _close(blk["DCT Complexity"], want[1])


def test_config_keys_pixfmt_and_device(tmp_path, y4m_case):
    from rtvqa_amd import _native as N
    from rtvqa_amd import video_processing as vp
    c = y4m_case
    cfg = {"crf": 23, "resize_width": 64, "resize_height": 64, "frame_interval": 5}
    out = str(tmp_path / "p.csv")
    # pixfmt names the layout of array inputs; the planar pair needs the BGR stream beside it, and the right plane sizes
    m = vp.process_video_and_extract_metrics(c["yr"], c["yd"], dict(cfg, pixfmt="yuv420p", device=0), csv_file=out, encoded_bgr=c["enc"])
    g = vp.process_video_and_extract_metrics(c["ref"], c["enc"], dict(cfg, pixfmt="bgr24"), csv_file=out)
    assert m["PSNR"] != g["PSNR"] and all(m[k] == g[k] for k in FIXED if m[k] == m[k])   # other planes, the same complexity half
    gray_r, gray_d = np.ascontiguousarray(c["ref"][..., 1]), np.ascontiguousarray(c["enc"][..., 1])
    y = vp.process_video_and_extract_metrics(gray_r, gray_d, dict(cfg, pixfmt="gray"), csv_file=out, encoded_bgr=c["enc"])
    lines = []
    vp.frame_quality(c["ref"], c["enc"], on_chunk=lambda n0, sse, ssim: lines.append(sse[0]))
    assert y["PSNR"] == float("%.2f" % (10 * np.log10(255.0 ** 2 / (float(lines[0][1]) / (c["h"] * c["w"])))))   # the G plane alone
    with pytest.raises(ValueError, match="encoded_bgr"):
        vp.process_video_and_extract_metrics(c["pr"], c["pd"], cfg, csv_file=out)
    with pytest.raises(ValueError, match="planar"):
        vp.process_video_and_extract_metrics(c["ref"], c["enc"], dict(cfg, pixfmt="yuv420p"), csv_file=out, encoded_bgr=c["enc"])
    with pytest.raises(ValueError, match="planar"):
        vp.process_video_and_extract_metrics(c["pr"], c["enc"], cfg, csv_file=out, encoded_bgr=c["enc"])
    with pytest.raises(ValueError, match="pixel layout"):
        vp.process_video_and_extract_metrics(c["ref"], c["pd"], cfg, csv_file=out, encoded_bgr=c["enc"])
    with pytest.raises(ValueError, match="same number of frames"):
        vp.process_video_and_extract_metrics(c["pr"], c["pd"], cfg, csv_file=out, encoded_bgr=c["enc"][:-1])
    with pytest.raises(N.VqaError):
        vp.process_video_and_extract_metrics(c["ref"], c["enc"], dict(cfg, device=63), csv_file=out)
    with pytest.raises(ValueError, match="share a geometry"):   # an equal-byte reshape is not the same clip
        vp.process_video_and_extract_metrics(c["ref"].reshape(c["n"], c["w"], c["h"], 3), c["enc"], cfg, csv_file=out)
    assert len(list(csv.reader(open(out)))) == 4   # one header, the three rows that succeeded


def test_two_threads_share_a_device(tmp_path):
    """The reference's surface is written for threaded callers (video_processing.py:25-41, :56-67): two threads, each on its own
    clip of a different geometry (the ring, the lane buffers and the scratch regrow under contention), get the rows the serial
    calls return, and the CSV they share has one header."""
    from rtvqa_amd import complexity_metrics as cm
    from rtvqa_amd import synth
    from rtvqa_amd import video_processing as vp
    clips = [_clip(40, 120, 160, seed=41), _clip(23, 144, 256, seed=42), _clip(31, 90, 130, seed=43)]
    encs = [synth.distort(c) for c in clips]
    cfgs = [{"crf": 23, "resize_width": 64, "resize_height": 64, "frame_interval": 3, "batch_size": 6},
            {"crf": 24, "resize_width": 48, "resize_height": 40, "frame_interval": 2, "batch_size": 4, "ssim_mode": "ffmpeg"},
            {"crf": 25, "resize_width": 130, "resize_height": 90, "frame_interval": 1, "batch_size": 9, "dct_mode": "full"}]
    serial = [vp.process_video_and_extract_metrics(c, e, g, csv_file=str(tmp_path / "s.csv")) for c, e, g in zip(clips, encs, cfgs)]
    cm.release_buffers()
    out = str(tmp_path / "t.csv")
    got, errs = [[None] * 4 for _ in clips], []

    def loop(k):
        try:
            for it in range(4):
                got[k][it] = vp.process_video_and_extract_metrics(clips[k], encs[k], cfgs[k], csv_file=out)
                if k == 0 and it == 1:
                    cm.release_buffers()      # (waits for the pass that is running, then the next one re-grows)
                if k == 1:
                    fr = clips[1]
                    assert cm.process_edge_frame(fr[it], 64, 64) == pl.process_edge_frame(fr[it], 64, 64)
        except BaseException as e:  # noqa: BLE001 - surfaced below
            errs.append((k, repr(e)))

    ts = [threading.Thread(target=loop, args=(k,)) for k in range(len(clips))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    for k in range(len(clips)):
        for it in range(4):
            assert got[k][it] == serial[k] or all(
                (a == b) or (a != a and b != b) for a, b in zip(got[k][it].values(), serial[k].values())), (k, it)
    rows = list(csv.reader(open(out)))
    assert len(rows) == 1 + 4 * len(clips) and sum(r[0] == "Bitrate (kbps)" for r in rows) == 1


def test_command_line_with_a_y4m_pair_and_the_encoded_bgr_stream(y4m_case, tmp_path, capsys):
    """video_processing.py:300-321's CLI with decoded streams: config.json (the reference's keys + ssim_mode / motion), the .y4m
    quality pair and --encoded-bgr; the row lands in the CSV and equals the function call's."""
    import json
    from rtvqa_amd import video_processing as vp
    c = y4m_case
    cfg = {"crf": 28, "vmaf_model_path": None, "resize_width": 64, "resize_height": 64, "frame_interval": 4, "ssim_mode": "ffmpeg",
           "motion": "farneback"}
    cfg_path = tmp_path / "config.json"
    cfg_path.write_text(json.dumps(cfg))
    out = str(tmp_path / "cli.csv")
    assert vp.main([str(cfg_path), c["pr"], c["pd"], "--encoded-bgr", c["pe"], "--csv", out]) == 0
    rows = list(csv.reader(open(out)))
    assert len(rows) == 2 and rows[1][rows[0].index("CRF")] == "28"
    want = vp.process_video_and_extract_metrics(c["pr"], c["pd"], cfg, csv_file=str(tmp_path / "f.csv"), encoded_bgr=c["enc"])
    got = dict(zip(rows[0], rows[1]))
    for k in ("PSNR", "SSIM", "Advanced Motion Complexity", "DCT Complexity", "Edge Detection Complexity"):
        assert float(got[k]) == float(want[k]), k
    # a torch tensor on the GPU as the encoded stream and pinned torch tensors as the planar pair: in place / DMA'd from
    import torch
    enc_t = torch.from_numpy(c["enc"]).cuda()
    yr_t, yd_t = torch.from_numpy(c["yr"]).pin_memory(), torch.from_numpy(c["yd"]).pin_memory()
    t = vp.process_video_and_extract_metrics(yr_t, yd_t, dict(cfg, pixfmt="yuv420p"), csv_file=str(tmp_path / "f.csv"), encoded_bgr=enc_t)
    for k in FIXED + ("PSNR", "SSIM"):
        assert t[k] == want[k] or (t[k] != t[k] and want[k] != want[k]), k
    bad = tmp_path / "bad.json"
    bad.write_text(json.dumps(dict(cfg, ssim_mode="psnr-hvs")))
    with pytest.raises(ValueError, match="ssim_mode must be"):
        vp.main([str(bad), c["pr"], c["pd"], "--encoded-bgr", c["pe"]])


def test_headerless_inputs_through_the_entry_point(y4m_case, tmp_path):
    """raw planar .yuv pair + raw packed .bgr24 stream, geometry from the config's height / width keys: the row of the .y4m / .npy call"""
    from rtvqa_amd import video_processing as vp
    c = y4m_case
    pr, pd_, pe = str(tmp_path / "r.yuv"), str(tmp_path / "d.yuv"), str(tmp_path / "enc.bgr24")
    c["yr"].tofile(pr)
    c["yd"].tofile(pd_)
    c["enc"].tofile(pe)
    cfg = {"crf": 23, "resize_width": 64, "resize_height": 64, "frame_interval": 4, "ssim_mode": "ffmpeg", "height": c["h"], "width": c["w"]}
    out = str(tmp_path / "raw.csv")
    raw = vp.process_video_and_extract_metrics(pr, pd_, cfg, csv_file=out, encoded_bgr=pe)
    ref = vp.process_video_and_extract_metrics(c["pr"], c["pd"], cfg, csv_file=out, encoded_bgr=c["pe"])
    for k in FIXED + ("PSNR", "SSIM", "Resolution (px)"):
        assert raw[k] == ref[k] or (raw[k] != raw[k] and ref[k] != ref[k]), k
    bgr_pair = vp.process_video_and_extract_metrics(pe, pe, cfg, csv_file=out)            # a raw BGR pair: the shared pass
    assert "PSNR" not in bgr_pair and bgr_pair["DCT Complexity"] == ref["DCT Complexity"]   # identical streams: psnr inf, no match (:160)
