"""The reference-shaped Python surface on the GPU against the oracle's reference-shaped pipeline."""
import functools
import re

import numpy as np
import pytest

from oracle import pipeline as pl

pytestmark = pytest.mark.gpu
RTOL = 1e-4


def _clip(n, h, w, seed=0):
    from rtvqa_amd import synth
    return synth.s_natural(n, h, w, seed=seed)


def _close(a, b, exact=False):
    if exact:
        return a == b
    return abs(float(a) - float(b)) <= RTOL * max(abs(float(b)), 1e-12)


def test_per_frame_callables_match_reference_shaped_oracle():
    from rtvqa_amd import complexity_metrics as cm
    fr = _clip(3, 120, 160, seed=1)
    f, p = fr[1], fr[0]
    assert _close(cm.process_dct_frame(f, 64, 64), pl.process_dct_frame(f, 64, 64, "full"))
    assert cm.process_histogram_frame(f, 64, 64) == pl.process_histogram_frame(f, 64, 64)
    assert cm.process_color_histogram_frame(f, 64, 64) == pl.process_color_histogram_frame(f, 64, 64)
    assert cm.process_edge_frame(f, 64, 64) == pl.process_edge_frame(f, 64, 64)
    assert isinstance(cm.process_edge_frame(f, 64, 64), np.int64)
    assert cm.process_frame_complexity((f, p)) == pl.process_frame_complexity((f, p))
    assert cm.process_frame_complexity((f, None)) == 0.0
    assert cm.process_orb_frame_for_parallel(f) == pl.process_orb_frame_for_parallel(f)
    spot = np.full((96, 128, 3), 50, np.uint8)
    spot[48:50, 64:66] = 250
    assert cm.process_orb_frame_for_parallel(spot) == pl.process_orb_frame_for_parallel(spot) == 1
    from oracle import c_oracle as co
    pg, cg = co.resize_linear(co.bgr2gray(p), 64, 64), co.resize_linear(co.bgr2gray(f), 64, 64)
    assert _close(cm.process_temporal_dct_frame(pg, cg, 64, 64), pl.process_temporal_dct_frame(pg, cg, 64, 64, "full"))


def test_process_in_batches_known_kernels_and_order():
    from rtvqa_amd import complexity_metrics as cm
    fr = list(_clip(7, 72, 96, seed=2))
    got = cm.process_in_batches(fr, functools.partial(cm.process_edge_frame, resize_width=48, resize_height=40), 4,
                                batch_size=3)
    want = [pl.process_edge_frame(f, 48, 40) for f in fr]
    assert got == want
    got = cm.process_in_batches(fr, cm.process_histogram_frame, 4, batch_size=100, resize_width=96, resize_height=72)
    assert got == [pl.process_histogram_frame(f, 96, 72) for f in fr]
    pairs = [(fr[i], fr[i - 1]) for i in range(1, 7)]
    got = cm.process_in_batches(pairs, cm.process_frame_complexity, 2, batch_size=4)
    assert got == [pl.process_frame_complexity(pr) for pr in pairs]
    assert cm.process_in_batches(fr, cm.process_orb_frame_for_parallel, 4, batch_size=3) == \
        [pl.process_orb_frame_for_parallel(f) for f in fr]
    unchained = [(fr[0], fr[3]), (fr[5], None), (fr[2], fr[6])]
    got = cm.process_in_batches(unchained, cm.process_frame_complexity, 2)
    assert got == [pl.process_frame_complexity(pr) for pr in unchained]


def test_process_in_batches_gray_and_mixed_size_items():
    """The batched path takes what the per-frame callables (and the reference's per-item executor.map, :146-147) take:
    2-D gray frames, and chunks whose frames differ in size; anything but uint8 is rejected, not silently cast."""
    from oracle import c_oracle as co
    from rtvqa_amd import complexity_metrics as cm
    fr = list(_clip(5, 72, 96, seed=4))
    gray = [co.bgr2gray(f) for f in fr]
    got = cm.process_in_batches(gray, cm.process_edge_frame, 2, batch_size=3, resize_width=48, resize_height=40)
    assert got == [cm.process_edge_frame(g, 48, 40) for g in gray]
    assert got == [pl.process_edge_frame(np.repeat(g[..., None], 3, 2), 48, 40) for g in gray]
    mixed = [fr[0], fr[1][:60, :80], gray[2], fr[3][:, :50], fr[4]]
    got = cm.process_in_batches(mixed, cm.process_histogram_frame, 2, batch_size=4, resize_width=32, resize_height=32)
    assert got == [cm.process_histogram_frame(np.ascontiguousarray(m), 32, 32) for m in mixed]
    with pytest.raises(ValueError):
        cm.process_in_batches([f.astype(np.float32) / 255 for f in fr], cm.process_dct_frame, 2, resize_width=32, resize_height=32)
    with pytest.raises(ValueError):
        cm.process_dct_frame(fr[0].astype(np.float32) / 255, 32, 32)


@pytest.mark.parametrize("interval,resize,n", [(10, (64, 64), 45), (1, (160, 120), 6), (10, (64, 64), 15)])
def test_calculate_average_scene_complexity(interval, resize, n):
    """config.json's configuration (64x64, interval 10) and a native-size interval-1 run."""
    from rtvqa_amd import complexity_metrics as cm
    fr = _clip(n, 120, 160, seed=3)
    native = resize == (160, 120)
    got = cm.calculate_average_scene_complexity(fr, resize[0], resize[1], frame_interval=interval, batch_size=2)
    want = pl.calculate_average_scene_complexity(list(fr), resize[0], resize[1], frame_interval=interval,
                                                 dct_mode="block8" if native else "full")
    assert len(got) == 8
    for k, (g, w_) in enumerate(zip(got, want)):
        if isinstance(w_, float) and np.isnan(w_):
            assert np.isnan(g), k
        else:
            assert _close(g, w_, exact=k in (3,)), (k, g, w_)


def test_run_ffmpeg_metrics_files(tmp_path):
    from rtvqa_amd import synth
    from rtvqa_amd import video_processing as vp
    from rtvqa_amd.engine import bgr_planes
    ref = _clip(3, 72, 104, seed=4)
    dist = synth.distort(ref)
    dist[2] = ref[2]  # identical frame: PSNR inf
    pl_, sl_ = str(tmp_path / "psnr.log"), str(tmp_path / "ssim.log")
    assert vp.run_ffmpeg_metrics(ref, dist, pl_, sl_, str(tmp_path / "vmaf.json")) is None
    m = vp.extract_metrics_from_logs(pl_, sl_, str(tmp_path / "vmaf.json"), "x", 23, 1000, "104x72", 30.0)
    sse, ssim = pl.frame_quality(ref[0], dist[0], bgr_planes(72, 104), "gauss")
    mse_avg = sum(sse) / (3.0 * 72 * 104)
    assert m["PSNR"] == pytest.approx(10 * np.log10(255 ** 2 / mse_avg), abs=6e-3)   # 2-decimal text
    assert m["SSIM"] == pytest.approx(sum(ssim) / 3, abs=2e-6)
    assert "VMAF" not in m
    lines = open(pl_).read().splitlines()
    assert len(lines) == 3 and lines[2].startswith("n:3 mse_avg:0.00") and "psnr_avg:inf" in lines[2]
    assert re.match(r"n:1 mse_avg:\d+\.\d\d mse_r:\d+\.\d\d mse_g:\d+\.\d\d mse_b:\d+\.\d\d psnr_avg:\d+\.\d\d ", lines[0])


def test_y4m_quality_and_pipeline_row(tmp_path):
    """yuv420p streams from .y4m files (the planes FFmpeg would compare) and the CSV row of
    process_video_and_extract_metrics, with the reference's label shift reproduced or fixed."""
    import csv
    from rtvqa_amd import frames, synth
    from rtvqa_amd import video_processing as vp
    from rtvqa_amd.engine import yuv420p_planes
    h, w = 72, 104
    ref = _clip(25, h, w, seed=6)
    enc = synth.distort(ref)
    yr, yd = frames.bgr_to_yuv420p(ref[:3]), frames.bgr_to_yuv420p(enc[:3])
    pr, pd_ = str(tmp_path / "r.y4m"), str(tmp_path / "d.y4m")
    frames.write_y4m(pr, yr, h, w)
    frames.write_y4m(pd_, yd, h, w)
    pl_, sl_ = str(tmp_path / "p.log"), str(tmp_path / "s.log")
    vp.run_ffmpeg_metrics(pr, pd_, pl_, sl_, str(tmp_path / "v.json"), ssim_mode="ffmpeg")
    line = open(sl_).read().splitlines()[0]
    sse, ssim = pl.frame_quality(yr[0], yd[0], yuv420p_planes(h, w), "ffmpeg")
    assert line.startswith("n:1 Y:%f U:%f V:%f All:" % tuple(ssim))
    assert open(pl_).read().startswith("n:1 mse_avg:%0.2f mse_y:%0.2f" % (sum(sse) / (1.5 * h * w), sse[0] / (h * w)))
    cfg = {"crf": 23, "resize_width": 64, "resize_height": 64, "frame_interval": 10}
    m_ref = vp.process_video_and_extract_metrics(ref, enc, cfg, csv_file=str(tmp_path / "a.csv"))
    m_fix = vp.process_video_and_extract_metrics(ref, enc, cfg, csv_file=str(tmp_path / "a.csv"), column_order="fixed")
    want = pl.calculate_average_scene_complexity(list(enc), 64, 64, frame_interval=10, dct_mode="full")
    assert m_fix["Histogram Complexity"] == pytest.approx(float(want[2]), rel=1e-6)
    assert m_ref["Temporal DCT Complexity"] == m_fix["Histogram Complexity"]       # the reference's shifted label
    assert m_fix["Temporal DCT Complexity"] == pytest.approx(float(want[6]), rel=RTOL)
    assert m_ref["Resolution (px)"] == "104x72" and "PSNR" in m_ref and "SSIM" in m_ref
    rows = list(csv.reader(open(str(tmp_path / "a.csv"))))
    assert len(rows) == 3 and rows[0][0] == "Bitrate (kbps)"


def test_device_resident_clip_with_interval():
    """A clip that already lives in HBM, frame_interval > 1: the selected frames are a strided view
    (frame_stride = interval * frame bytes), no gather copy; same tuple as the host path."""
    from rtvqa_amd import complexity_metrics as cm
    fr = _clip(37, 96, 128, seed=7)
    dev = cm.get_engine().upload(fr)
    a = cm.calculate_average_scene_complexity(fr, 64, 48, frame_interval=5, batch_size=3)
    b = cm.calculate_average_scene_complexity(dev, 64, 48, frame_interval=5, batch_size=3)
    for x, y in zip(a, b):
        assert (np.isnan(x) and np.isnan(y)) or x == y
    want = pl.calculate_average_scene_complexity(list(fr), 64, 48, frame_interval=5, dct_mode="full")
    assert _close(a[1], want[1]) and a[3] == want[3] and _close(a[6], want[6]) and a[0] == want[0]


def _sharded_gpu_worker(rank, world, port, clip, interval, resize, out_path, backend="gloo"):
    import os
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["VQA_DEVICE"] = "0"  # a 1-GPU box: every rank on device 0 (on a node: LOCAL_RANK)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from rtvqa_amd import complexity_metrics as cm
    got = cm.calculate_average_scene_complexity_sharded(clip, resize[0], resize[1], frame_interval=interval, batch_size=4)
    np.save(out_path % rank, np.array(got, np.float64))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n,interval,resize,world,backend", [(64, 5, (64, 64), 2, "gloo"), (23, 2, (80, 60), 3, "gloo"),
                                                            (23, 2, (80, 60), 1, "nccl")])
def test_sharded_stream_on_gpu_equals_single_process(tmp_path, n, interval, resize, world, backend):
    """one long stream split over ranks (contiguous ranges + 1-frame halo), every rank running
    the real HIP path, pooled by one scalar all-reduce — equals the single-process 8-tuple."""
    import socket
    import torch.multiprocessing as mp
    from rtvqa_amd import complexity_metrics as cm
    clip = _clip(n, 120, 160, seed=9)
    want = np.array(cm.calculate_average_scene_complexity(clip, resize[0], resize[1], frame_interval=interval), np.float64)
    with socket.socket() as sck:
        sck.bind(("127.0.0.1", 0))
        port = sck.getsockname()[1]
    out = str(tmp_path / "g%d.npy")
    # (RCCL refuses two ranks on one device, so the device-tensor reduction is exercised with one rank here)
    mp.spawn(_sharded_gpu_worker, args=(world, port, clip, interval, resize, out, backend), nprocs=world, join=True)
    for r in range(world):
        got = np.load(out % r)
        for k in range(8):
            assert abs(got[k] - want[k]) <= 1e-12 * max(abs(want[k]), 1e-30), (r, k, got[k], want[k])


def test_host_clip_batches_pipeline_over_two_engines():
    """Host clips longer than one batch ping-pong over two engines (copy/compute overlap): same numbers,
    same order as the one-engine path."""
    from rtvqa_amd import complexity_metrics as cm
    from rtvqa_amd import video_processing as vp
    from rtvqa_amd import synth
    clip = _clip(40, 120, 160, seed=12)
    one = cm.complexity_series(clip, 64, 64, frame_interval=1, batch_size=7, engine=cm.get_engine())
    two = cm.complexity_series(clip, 64, 64, frame_interval=1, batch_size=7)
    big = cm.complexity_series(clip, 64, 64, frame_interval=1, batch_size=100)
    for k in ("motion", "dct", "hist", "edge", "orb", "color", "temporal"):
        assert len(one[k]) == len(two[k]) == len(big[k]) == (38 if k == "temporal" else 39)
        assert all(a == b == c for a, b, c in zip(one[k], two[k], big[k])), k
    dist = synth.distort(clip)
    s1 = vp.frame_quality(clip, dist, "bgr24", "gauss", batch_size=6, engine=cm.get_engine())
    s2 = vp.frame_quality(clip, dist, "bgr24", "gauss", batch_size=6)
    assert (s1[0] == s2[0]).all() and (s1[1] == s2[1]).all() and s1[0].shape == (40, 3)


def test_two_engines_from_two_threads():
    """One ctx per host thread (include/vqa.h): two threads, each with its own engine on the same device,
    running different work concurrently, get the results a single thread gets."""
    import threading
    import rtvqa_amd
    from rtvqa_amd import _native as N
    from rtvqa_amd import synth
    from rtvqa_amd.engine import bgr_planes
    clips = [_clip(9, 144, 256, seed=31), _clip(9, 90, 130, seed=32)]
    dists = [synth.distort(c) for c in clips]

    def work(eng, k):
        c, d = clips[k], dists[k]
        rec = eng.complexity(d[1:], prev0=d[0], mask=N.M_ALL, dct_mode=N.DCT_BLOCK8)
        # field by field (the records carry alignment padding); hyst_steps is a diagnostic that depends on scheduling
        q = eng.quality(c, d, bgr_planes(c.shape[1], c.shape[2]), N.SSIM_GAUSS)
        return (tuple(rec[f].tobytes() for f in rec.dtype.names if f != "hyst_steps"),
                tuple(q[f].tobytes() for f in q.dtype.names))

    with rtvqa_amd.Engine(0) as e0, rtvqa_amd.Engine(0) as e1:
        want = [work(e0, 0), work(e0, 1)]
        got = [[None] * 6, [None] * 6]

        def loop(eng, k):
            for it in range(6):
                got[k][it] = work(eng, k)

        ts = [threading.Thread(target=loop, args=(e, k)) for k, e in enumerate((e0, e1))]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
    for k in range(2):
        for it, g in enumerate(got[k]):
            assert g == want[k], "thread %d iteration %d differs" % (k, it)


def test_farneback_motion_mode_through_the_reference_surface():
    """set_motion_mode("farneback"): process_frame_complexity and slot 0 of the 8-tuple become the
    reference's own Farneback metric (:340-343)."""
    from rtvqa_amd import complexity_metrics as cm
    clip = _clip(24, 120, 160, seed=14)
    cm.set_motion_mode("farneback")
    try:
        f, p = clip[5], clip[4]
        got = cm.process_frame_complexity((f, p))
        assert isinstance(got, np.float32) and _close(got, pl.process_frame_complexity((f, p), motion="farneback"))
        pairs = [(clip[i], clip[i - 1]) for i in range(1, 6)]
        got = cm.process_in_batches(pairs, cm.process_frame_complexity, 2, batch_size=3)
        want = [pl.process_frame_complexity(pr, motion="farneback") for pr in pairs]
        assert all(_close(a, b) for a, b in zip(got, want))
        t = cm.calculate_average_scene_complexity(clip, 64, 64, frame_interval=4, batch_size=3)
        w_ = pl.calculate_average_scene_complexity(list(clip), 64, 64, frame_interval=4, motion="farneback")
        assert _close(t[0], w_[0]) and all(_close(a, b, exact=k == 3) for k, (a, b) in enumerate(zip(t[1:], w_[1:]), 1))
    finally:
        cm.set_motion_mode("sad")
    assert cm.process_frame_complexity((clip[5], clip[4])) == pl.process_frame_complexity((clip[5], clip[4]))
