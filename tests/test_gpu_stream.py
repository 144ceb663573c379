"""stream.run - the one-pass host side of the reference's entry points - on the GPU: wherever the frames live
(pageable NumPy, np.load(mmap_mode="r"), pinned, device) and however the clip is chunked, the series and the
quality numbers are the same bits; the fused pass (process_video_and_extract_metrics: one upload for both halves)
equals the two separate passes; vqa_trim gives the memory back and changes no result; the table caches stay bounded."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import c_oracle as co
from oracle import pipeline as pl

pytestmark = pytest.mark.gpu


def _clip(n, h, w, seed=0):
    from rtvqa_amd import synth
    return synth.s_natural(n, h, w, seed=seed)


def _same_series(a, b):
    for k in ("motion", "dct", "hist", "edge", "orb", "color", "temporal"):
        assert len(a[k]) == len(b[k]), k
        assert all((x == y) or (x != x and y != y) for x, y in zip(a[k], b[k])), k
    assert a["range"] == b["range"]


@pytest.mark.parametrize("interval,batch", [(1, 7), (3, 4), (10, 100)])
def test_every_residence_gives_the_same_bits(tmp_path, interval, batch):
    from rtvqa_amd import complexity_metrics as cm
    from rtvqa_amd import stream
    from rtvqa_amd import video_processing as vp
    from rtvqa_amd import synth
    n, h, w = 43, 120, 168
    ref = _clip(n, h, w, seed=50 + interval)
    dist = synth.distort(ref)
    eng = cm.get_engine()
    assert not eng.is_pinned(dist)
    pin_r, pin_d = eng.alloc_pinned(ref.shape), eng.alloc_pinned(dist.shape)
    pin_r[...] = ref
    pin_d[...] = dist
    assert eng.is_pinned(pin_d) and eng.is_pinned(pin_d[5:])
    np.save(str(tmp_path / "d.npy"), dist)
    np.save(str(tmp_path / "r.npy"), ref)
    dev_r, dev_d = eng.upload(ref), eng.upload(dist)
    assert not eng.is_pinned(np.zeros(4, np.uint8))
    want = cm.complexity_series(dist, 64, 48, interval, batch_size=1000, engine=eng)   # one chunk, one engine
    for src in (dist, pin_d, str(tmp_path / "d.npy"), dev_d, dist[:, :, :, :]):
        _same_series(cm.complexity_series(src, 64, 48, interval, batch_size=batch), want)
    # the reference-shaped oracle pipeline on the same clip (counts exact, floats 1e-4)
    t = cm.calculate_average_scene_complexity(pin_d, 64, 48, frame_interval=interval, batch_size=batch)
    o = pl.calculate_average_scene_complexity(list(dist), 64, 48, frame_interval=interval, dct_mode="full")
    for k in (2, 3, 4, 5):
        assert (np.isnan(t[k]) and np.isnan(o[k])) or float(t[k]) == pytest.approx(float(o[k]), rel=1e-12), k
    for k in (0, 1, 6):
        assert (np.isnan(t[k]) and np.isnan(o[k])) or float(t[k]) == pytest.approx(float(o[k]), rel=1e-4), k
    q_want = vp.frame_quality(ref, dist, batch_size=1000, engine=eng)
    for r, d in ((ref, dist), (pin_r, pin_d), (dev_r, dev_d), (pin_r, dist),
                 (np.load(str(tmp_path / "r.npy"), mmap_mode="r"), np.load(str(tmp_path / "d.npy"), mmap_mode="r"))):
        q = vp.frame_quality(r, d, batch_size=batch)
        assert np.array_equal(q[0], q_want[0]) and np.array_equal(q[1], q_want[1])
    # the fused pass: both halves from one upload
    for r, d in ((ref, dist), (pin_r, pin_d), (dev_r, dev_d)):
        got_q, got_s = stream.run(d, r, quality=stream.Quality(vp.bgr_planes(h, w)),
                                  complexity=stream.Complexity((64, 48), interval), batch_size=batch)
        assert np.array_equal(got_q[0], q_want[0]) and np.array_equal(got_q[1], q_want[1])
        _same_series(got_s, want)
    eng.free_pinned(pin_r)
    eng.free_pinned(pin_d)


def test_region_of_interest_and_short_clips_through_the_ring():
    from rtvqa_amd import complexity_metrics as cm
    big = _clip(9, 100, 140, seed=77)
    roi = big[:, 3:83, 5:133]                       # padded rows: the ring compacts them
    want = cm.complexity_series(np.ascontiguousarray(roi), 64, 64, 2, batch_size=100)
    _same_series(cm.complexity_series(roi, 64, 64, 2, batch_size=2), want)
    assert cm.complexity_series(big[:1], 64, 64, 1)["dct"] == []          # nothing to measure: no engine work
    one = cm.complexity_series(big[:2], 64, 64, 1)
    assert len(one["dct"]) == 1 and one["temporal"] == []


def test_pipeline_row_is_one_pass_and_matches_the_two_calls(tmp_path):
    """process_video_and_extract_metrics (video_processing.py:216 + :242 on the same encoded stream) = the numbers of
    run_ffmpeg_metrics + calculate_average_scene_complexity called one after the other."""
    from rtvqa_amd import complexity_metrics as cm
    from rtvqa_amd import synth
    from rtvqa_amd import video_processing as vp
    ref = _clip(33, 96, 128, seed=5)
    enc = synth.distort(ref)
    cfg = {"crf": 23, "resize_width": 64, "resize_height": 64, "frame_interval": 4, "batch_size": 5}
    m = vp.process_video_and_extract_metrics(ref, enc, cfg, csv_file=str(tmp_path / "a.csv"), column_order="fixed")
    pl_, sl_ = str(tmp_path / "p.log"), str(tmp_path / "s.log")
    vp.run_ffmpeg_metrics(ref, enc, pl_, sl_, str(tmp_path / "v.json"))
    m2 = vp.extract_metrics_from_logs(pl_, sl_, str(tmp_path / "v.json"), "x", 23, 0, "128x96", 30.0)
    assert m["PSNR"] == m2["PSNR"] and m["SSIM"] == m2["SSIM"]
    assert len(open(pl_).read().splitlines()) == 33
    t = cm.calculate_average_scene_complexity(enc, 64, 64, frame_interval=4)
    names = ("Advanced Motion Complexity", "DCT Complexity", "Histogram Complexity", "Edge Detection Complexity",
             "ORB Feature Complexity", "Color Histogram Complexity", "Temporal DCT Complexity", "Framerate Variation")
    for name, v in zip(names, t):
        assert m[name] == v, name


def _free_bytes():
    import torch
    return torch.cuda.mem_get_info(0)[0]


def test_trim_gives_the_memory_back_and_changes_nothing():
    """vqa_trim after a 2160p Farneback submit: the device's free memory returns to within 64 MiB of what it was
    before the submit, a pending batch refuses the trim, and the same ctx then returns oracle-exact records."""
    import rtvqa_amd
    from rtvqa_amd import _native as N
    with rtvqa_amd.Engine(0) as eng:
        small = _clip(3, 144, 256, seed=9)
        first = eng.complexity(small[1:], prev0=small[0], mask=N.M_ALL, dct_mode=N.DCT_BLOCK8)
        eng.trim()
        base = _free_bytes()
        big = eng.upload(_clip(3, 2160, 3840, seed=10))
        held = _free_bytes()
        eng.complexity_submit(big.slice(1, 3), big.frame(0), N.M_ALL, eng.make_params(motion_mode=N.MOTION_FARNEBACK, dct_mode=N.DCT_FULL))
        assert eng.lib.vqa_trim(eng.ctx) == N.VQA_ERR_STATE          # a batch is pending
        rec = eng.complexity_wait()
        assert rec[0]["flow_mag_mean"] > 0
        grown = _free_bytes()
        assert held - grown > (1 << 30)                                # GiB-sized scratch is what trim is for
        eng.trim()
        big._owner.free()
        after = _free_bytes()
        assert abs(after - base) <= (64 << 20), (base, held, grown, after)
        again = eng.complexity(small[1:], prev0=small[0], mask=N.M_ALL, dct_mode=N.DCT_BLOCK8)
        for f in first.dtype.names:
            if f != "hyst_steps":
                assert first[f].tobytes() == again[f].tobytes(), f
        g, gp = co.bgr2gray(small[1]), co.bgr2gray(small[0])
        assert int(again[0]["edge_count"]) == co.canny(g, 100, 200)[0]
        assert (again[0]["hist_gray"] == co.hist_u8(g)).all()
        assert int(again[0]["sad_sum"]) == co.block_sad(gp, g, 7)[1]


def test_table_caches_stay_bounded_over_200_geometries():
    """Every distinct (frame size, resize) pair adds device tables; beyond VQA_TABLE_CACHE_GEOMETRIES the least recently
    used set goes.  200 geometries on one ctx: the device's free memory stays where it was after the first 20, and an
    evicted geometry still gives oracle-exact bins when it comes back."""
    import rtvqa_amd
    from rtvqa_amd import _native as N
    rng = np.random.default_rng(0)
    with rtvqa_amd.Engine(0) as eng:
        fr = rng.integers(0, 256, (2, 96, 160, 3), dtype=np.uint8)
        dev = eng.upload(fr)
        first = None
        mark = None
        for i in range(200):
            rw, rh = 130 + 2 * (i % 100), 128 + 2 * (i // 100)      # even, 2-3-5-smooth or not: FFT plans and dense matrices
            rec = eng.complexity(dev.slice(1, 2), prev0=dev.frame(0), mask=N.M_GRAY_HIST | N.M_DCT | N.M_TEMPORAL_DCT,
                                 resize=(rw, rh), dct_mode=N.DCT_FULL)
            if i == 0:
                first = rec.copy()
            if i == 20:
                mark = _free_bytes()
        assert mark - _free_bytes() <= (32 << 20)
        rec = eng.complexity(dev.slice(1, 2), prev0=dev.frame(0), mask=N.M_GRAY_HIST | N.M_DCT | N.M_TEMPORAL_DCT,
                             resize=(130, 128), dct_mode=N.DCT_FULL)
        assert rec["hist_gray"].tobytes() == first["hist_gray"].tobytes()
        assert rec["dct_energy"].tobytes() == first["dct_energy"].tobytes()
        g = co.bgr2gray(co.resize_linear(fr[1], 130, 128))
        assert (rec[0]["hist_gray"] == co.hist_u8(g)).all()


def test_release_buffers_then_the_api_still_works():
    """release_buffers(): the pinned ring, the lane buffers and (vqa_trim) every engine's scratch go back - the device's free
    memory returns to where it was before the passes (within 64 MiB) - and the next call re-grows what it needs."""
    from rtvqa_amd import complexity_metrics as cm
    from rtvqa_amd import synth
    from rtvqa_amd import video_processing as vp
    clip = _clip(12, 90, 120, seed=3)
    a = cm.complexity_series(clip, 64, 64, 1, batch_size=4)
    cm.release_buffers()
    base = _free_bytes()
    big = _clip(40, 540, 960, seed=4)                       # 62 MB per stream: lane buffers, ring and scratch well above the bar
    vp.frame_quality(big, synth.distort(big), batch_size=16)
    cm.complexity_series(big, 960, 540, 1, batch_size=16)
    assert base - _free_bytes() > (64 << 20)
    cm.release_buffers()
    assert abs(_free_bytes() - base) <= (64 << 20)
    _same_series(cm.complexity_series(clip, 64, 64, 1, batch_size=4), a)


def test_torch_tensors_are_accepted_in_place():
    """north_star: the host holds decoded frames in pinned buffers - PyTorch-ROCm tensors.  A CUDA(HIP) uint8 tensor is used in
    place (zero-copy DeviceFrames), a pin_memory CPU tensor is recognised as page-locked and DMA'd from directly; both give
    the bits of the NumPy clip."""
    import torch
    from rtvqa_amd import complexity_metrics as cm
    from rtvqa_amd import synth
    from rtvqa_amd import video_processing as vp
    ref = _clip(21, 96, 128, seed=61)
    dist = synth.distort(ref)
    want = cm.complexity_series(dist, 64, 64, 2, batch_size=4)
    t_pin = torch.from_numpy(dist).pin_memory()
    assert cm.get_engine().is_pinned(t_pin.numpy())
    t_dev = torch.from_numpy(dist).cuda()
    for src in (t_pin, t_dev, torch.from_numpy(dist)):
        _same_series(cm.complexity_series(src, 64, 64, 2, batch_size=4), want)
    t = cm.calculate_average_scene_complexity(t_dev, 64, 64, frame_interval=2)
    t2 = cm.calculate_average_scene_complexity(dist, 64, 64, frame_interval=2)
    assert all((a == b) or (a != a and b != b) for a, b in zip(t, t2))
    with pytest.raises(ValueError):
        cm.complexity_series(torch.zeros(3, 8, 8, 3), 8, 8, 1)
    m = vp.process_video_and_extract_metrics(torch.from_numpy(ref).cuda(), t_dev, {"resize_width": 64, "resize_height": 64, "frame_interval": 2},
                                             csv_file=os.devnull)
    m2 = vp.process_video_and_extract_metrics(ref, dist, {"resize_width": 64, "resize_height": 64, "frame_interval": 2}, csv_file=os.devnull)
    assert m["PSNR"] == m2["PSNR"] and m["SSIM"] == m2["SSIM"] and m["DCT Complexity"] == m2["DCT Complexity"]


def test_a_failure_inside_a_pass_leaves_the_engines_usable(tmp_path):
    """The reference's convention for a failed step is log, re-raise, nothing left running (video_processing.py:295-297).
    A callback that raises in the middle of a pass (here: the stats writer of chunk 1): the exception surfaces, nothing is
    pending on either lane afterwards, and the next pass on the same engines returns the right bits."""
    from rtvqa_amd import _native as N
    from rtvqa_amd import complexity_metrics as cm
    from rtvqa_amd import stream, synth
    from rtvqa_amd import video_processing as vp
    ref = _clip(30, 96, 128, seed=71)
    dist = synth.distort(ref)
    want_q, want_s = stream.run(dist, ref, stream.Quality(vp.bgr_planes(96, 128)), stream.Complexity((64, 64), 2), batch_size=1000,
                                engine=cm.get_engine())
    calls = []

    def boom(first, sse, ssim):
        calls.append(first)
        if len(calls) == 2:
            raise OSError("disk full")

    with pytest.raises(OSError):
        stream.run(dist, ref, stream.Quality(vp.bgr_planes(96, 128)), stream.Complexity((64, 64), 2), batch_size=4, on_quality=boom)
    for e in stream.get_engine_pair():
        buf = (N.VqaFrameMetrics * 1)()
        assert e.lib.vqa_complexity_wait(e.ctx, buf, 1) == N.VQA_ERR_STATE   # nothing is pending on the C side either
    got_q, got_s = stream.run(dist, ref, stream.Quality(vp.bgr_planes(96, 128)), stream.Complexity((64, 64), 2), batch_size=4)
    assert np.array_equal(got_q[0], want_q[0]) and np.array_equal(got_q[1], want_q[1])
    _same_series(got_s, want_s)
    # a submit that the C side refuses (a plane smaller than the SSIM window) surfaces as VqaError, same guarantees
    tiny = np.zeros((5, 8, 8, 3), np.uint8)
    with pytest.raises(N.VqaError):
        vp.frame_quality(tiny, tiny, batch_size=2)
    _same_series(stream.run(dist, complexity=stream.Complexity((64, 64), 2), batch_size=4)[1], want_s)


@pytest.mark.parametrize("lanes", [1, 3])
def test_any_number_of_lanes_gives_the_same_bits(lanes, monkeypatch):
    """stream.MAX_LANES engines alternate a pass's chunks (default 2).  One lane (everything on the default engine, in order)
    and three (a ring of four slots, three lane buffers) must return what two return."""
    from rtvqa_amd import stream, synth
    from rtvqa_amd import video_processing as vp
    ref = _clip(37, 90, 122, seed=81)
    dist = synth.distort(ref)
    q, cx = stream.Quality(vp.bgr_planes(90, 122)), stream.Complexity((64, 64), 3)
    want_q, want_s = stream.run(dist, ref, q, cx, batch_size=5)
    monkeypatch.setattr(stream, "MAX_LANES", lanes)
    dev_r, dev_d = stream.get_engine().upload(ref), stream.get_engine().upload(dist)
    for r, d in ((ref, dist), (dev_r, dev_d)):
        got_q, got_s = stream.run(d, r, q, cx, batch_size=5)
        assert np.array_equal(got_q[0], want_q[0]) and np.array_equal(got_q[1], want_q[1])
        _same_series(got_s, want_s)


@pytest.mark.parametrize("mode", ["gauss", "ffmpeg"])
def test_ssim_does_not_depend_on_the_batch_size(mode):
    """include/vqa.h, vqa_plane_metrics.ssim: at 1080p a launch of 3 frames cuts the planes into more row strips than a launch
    of 25 (the strip count follows the workgroup count).  Rounds 1-5 summed the SSIM map in floats per strip and the two
    differed in the last digits (1.4e-9); the map is now summed in 2^-27 fixed point - integer sums do not care where the
    strips are cut - so the same frame pair gives the same bits in any batch: the CSV row of a clip cannot change in its
    last printed digit with batch_size or with where the clip lives (host chunks are capped in bytes, resident ones are not)."""
    from rtvqa_amd import synth
    from rtvqa_amd import video_processing as vp
    ref = _clip(25, 1080, 1920, seed=91)
    dist = synth.distort(ref)
    a = vp.frame_quality(ref, dist, ssim_mode=mode, batch_size=25)
    b = vp.frame_quality(ref, dist, ssim_mode=mode, batch_size=3)
    c = vp.frame_quality(ref, dist, ssim_mode=mode, batch_size=7)
    one = vp.frame_quality(ref[11:12], dist[11:12], ssim_mode=mode)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[0], c[0])
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[1], c[1]) and np.array_equal(a[1][11], one[1][0])
    assert 0.5 < a[1].min() and a[1].max() < 1.0


def test_a_clip_that_only_starts_in_registered_memory_is_not_dma_ed_from():
    """vqa_host_is_pinned probes the first AND the last byte: a clip whose head lies in a hipHostRegister'ed region and whose
    tail does not is pageable as far as a whole-range DMA is concerned - it goes through the ring and gives the same bits."""
    import torch
    from rtvqa_amd import complexity_metrics as cm
    n, h, w = 12, 72, 96
    clip = _clip(n, h, w, seed=77)
    page = 4096
    raw = np.zeros(clip.nbytes + 2 * page, np.uint8)
    off = (-raw.ctypes.data) % page
    buf = raw[off:off + clip.nbytes].reshape(clip.shape)
    buf[...] = clip
    half = (clip.nbytes // 2) // page * page
    eng = cm.get_engine()
    rt = torch.cuda.cudart()
    assert int(rt.cudaHostRegister(buf.ctypes.data, half, 0)) == 0
    try:
        head = buf.reshape(-1)[:half]
        assert eng.is_pinned(head) and eng.is_pinned(head[100:200])
        assert not eng.is_pinned(buf) and not eng.is_pinned(buf.reshape(-1)[half - 10:half + 10])
        assert not eng.is_pinned(buf[::-1])
        _same_series(cm.complexity_series(buf, 64, 48, 2, batch_size=4), cm.complexity_series(clip, 64, 48, 2, batch_size=4))
    finally:
        assert int(rt.cudaHostUnregister(buf.ctypes.data)) == 0


def test_stream_wait_orders_a_measuring_context_behind_the_copy_lane():
    """vqa_stream_wait (include/vqa.h): uploads enqueued on one context (the copy lane), kernels on another that waits for
    them ON THE DEVICE.  The device buffer is overwritten with a different clip every round and measured at once, with no
    host wait in between: a kernel that started before its upload had landed would measure the previous clip."""
    import rtvqa_amd
    from rtvqa_amd import _native as N
    from rtvqa_amd.engine import DeviceBuffer, DeviceFrames
    n, h, w = 24, 360, 640
    clips = [_clip(n, h, w, seed=200 + k) for k in range(4)]
    with rtvqa_amd.Engine(0) as work, rtvqa_amd.Engine(0) as cp:
        want = [work.complexity(c[1:], prev0=c[0], mask=N.M_GRAY_HIST | N.M_EDGE | N.M_DCT, dct_mode=N.DCT_BLOCK8) for c in clips]
        pins = []
        for c in clips:
            p = cp.alloc_pinned(c.shape)
            p[...] = c
            pins.append(p)
        buf = DeviceBuffer(work, clips[0].nbytes)
        fr = DeviceFrames(buf.ptr, n, h, w, owner=buf)
        for rnd in range(12):
            k = rnd % len(clips)
            cp.h2d_async(buf.ptr, pins[k].ctypes.data, pins[k].nbytes)
            work.wait_for(cp)
            got = work.complexity(fr.slice(1, n), prev0=fr.frame(0), mask=N.M_GRAY_HIST | N.M_EDGE | N.M_DCT, dct_mode=N.DCT_BLOCK8)
            cp.wait_for(work)      # (and the other way round: the next upload must not overtake this round's kernels)
            for f in ("hist_gray", "edge_count", "dct_energy", "sum_gray2"):
                assert np.array_equal(got[f], want[k][f]), (rnd, f)
        st = work.lib.vqa_stream_wait(work.ctx, work.ctx)
        assert st == N.VQA_ERR_INVALID and work.lib.vqa_stream_wait(work.ctx, None) == N.VQA_ERR_INVALID
        for p in pins:
            cp.free_pinned(p)
        buf.free()
