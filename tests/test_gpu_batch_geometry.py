"""Parity at the launch geometry bench.py times.

The timed workloads are one device-resident batch of 256 x 1080p (BASELINE configs[1]/[2], "c2"/"c3") or 64 x 2160p
(configs[3], "c4") frame pairs.  At that size the kernels take paths a 2-4 frame batch never reaches: the marching
8x8 DCT runs 16-frame chunks with a halo frame per chunk and a grid-filling shrink rule, the Gaussian SSIM picks its
strip count from n, the Canny tail runs one persistent workgroup per frame, the SAD prefetch runs across frame seams.
These tests submit exactly those batches (same synthetic stream, same masks, B/G/R planes, both SSIM modes) and
compare a spread of batch positions - both ends, the middle and both sides of the 16-frame seams - with the oracle:
counts / SAD / SSE exact, DCT / SSIM to 1e-4.
"""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import bench
from oracle import check

pytestmark = pytest.mark.gpu


def _resident_stream(engine, h, w, B, keep, content="natural", rank=0):
    """Upload stream frames 0..B (B+1 frames: prev0 + the batch) as bench.py does; returns device views and host
    copies of the stream indices in `keep`."""
    from rtvqa_amd import _native as N
    from rtvqa_amd.engine import DeviceBuffer, DeviceFrames
    fb = h * w * 3
    ref_buf, dist_buf = DeviceBuffer(engine, fb * (B + 1)), DeviceBuffer(engine, fb * (B + 1))
    host = {}
    for a in range(0, B + 1, bench.CHUNK):
        n = min(bench.CHUNK, B + 1 - a)
        r, d = bench.stream_chunk(content, rank, h, w, a, n)
        N.check(engine.lib.vqa_copy_h2d(engine.ctx, ref_buf.ptr + a * fb, r.ctypes.data, r.nbytes), "h2d", engine.ctx)
        N.check(engine.lib.vqa_copy_h2d(engine.ctx, dist_buf.ptr + a * fb, d.ctypes.data, d.nbytes), "h2d", engine.ctx)
        engine.sync()
        for i in keep:
            if a <= i < a + n:
                host[i] = (r[i - a].copy(), d[i - a].copy())
    ref_all = DeviceFrames(ref_buf.ptr, B + 1, h, w, owner=ref_buf)
    dist_all = DeviceFrames(dist_buf.ptr, B + 1, h, w, owner=dist_buf)
    return ref_all, dist_all, host


def _run_and_check(engine, h, w, B, positions, content="natural"):
    from rtvqa_amd import _native as N
    from rtvqa_amd.engine import bgr_planes
    keep = sorted(set(positions) | set(j + 1 for j in positions))
    ref_all, dist_all, host = _resident_stream(engine, h, w, B, keep, content)
    ref_b, dist_b, prev0 = ref_all.slice(1, B + 1), dist_all.slice(1, B + 1), dist_all.frame(0)

    # the oracle runs on host threads (ctypes releases the GIL) while the GPU works
    def exp(j):
        return j, check.expected(host[j + 1][0], host[j + 1][1], host[j][1], True, ("gauss", "ffmpeg"))
    pool = ThreadPoolExecutor(8)
    fut = pool.map(exp, positions)

    params = engine.make_params(dct_mode=N.DCT_BLOCK8, motion_mode=N.MOTION_SAD)
    planes = bgr_planes(h, w)
    # exactly bench.py's step: quality and complexity submitted back to back on the context's stream
    engine.quality_submit(ref_b, dist_b, planes, N.SSIM_GAUSS)
    engine.complexity_submit(dist_b, prev0, N.M_ALL, params)
    qg = engine.quality_wait()
    c = engine.complexity_wait()
    qf = engine.quality(ref_b, dist_b, planes, N.SSIM_FFMPEG)
    # the c2 mask (DCT only) launches the marching kernel without the gray histograms around it
    c2 = engine.complexity(dist_b, prev0=prev0, mask=N.M_DCT | N.M_TEMPORAL_DCT, params=params)
    assert c.shape == (B,) and qg.shape == (B, 3) and not c["hyst_overflow"].any()
    assert (c2["dct_energy"] == c["dct_energy"]).all() and (c2["temporal_dct_l1"] == c["temporal_dct_l1"]).all()

    # size-independent properties over ALL frames of the batch
    assert (c["hist_gray"].sum(axis=1) == h * w).all() and (c["hist_bgr"].sum(axis=2) == h * w).all()
    rel = np.abs(c["dct_energy"] - c["sum_gray2"].astype(np.float64)) / c["sum_gray2"].astype(np.float64)
    assert rel.max() < check.RTOL                                  # Parseval, every frame
    assert (c["has_prev"] == 1).all() and (c["temporal_dct_l1"] > 0).all()
    assert (qg["sse"] == qf["sse"]).all()                          # the two SSIM kernels share the exact SSE
    assert (c["sad_blocks"] == (h // 16) * (w // 16)).all()
    assert (c["mv_d2_hist"].sum(axis=1) == c["sad_blocks"]).all()

    bad = {}
    for j, e in fut:
        m = check.compare(e, c[j], qg[j], "gauss") + check.compare(e, None, qf[j], "ffmpeg")
        if m:
            bad[j] = m
    pool.shutdown()
    assert not bad, bad


def test_c3_batch_256x1080p_at_the_timed_geometry(engine):
    """BASELINE configs[1] and [2] as bench.py launches them: 256 device-resident 1080p pairs, full suite, B/G/R planes."""
    _run_and_check(engine, 1080, 1920, 256, [0, 1, 15, 16, 17, 31, 32, 128, 255])


def test_c4_batch_64x2160p_at_the_timed_geometry(engine):
    """BASELINE configs[3] (and the per-GPU workload of configs[4]): 64 device-resident 2160p pairs, B/G/R planes."""
    _run_and_check(engine, 2160, 3840, 64, [0, 15, 16, 63])


def test_c3_noise_batch_seams(engine):
    """S-noise (worst case for the Canny fan-out and the histogram bins) at 1080p, 48 frames = three DCT chunks."""
    _run_and_check(engine, 1080, 1920, 48, [0, 15, 16, 47], content="noise")


def test_c3ref_batch_64x1080p_reference_definitions(engine):
    """`bench.py --workload c3ref` as a test: 64 device-resident 1080p pairs through the reference's OWN definitions -
    Farneback motion (one 64-pair chunk: the fused iteration runs 3 row strips x 8 column blocks per pair, the level
    kernels every tile shape) and the full-frame DCT (in-place FFT rows, eight-column slabs) - against the oracle at both
    ends and the middle of the batch, plus Parseval and positivity over every frame."""
    from rtvqa_amd import _native as N
    h, w, B, positions = 1080, 1920, 64, [0, 1, 32, 63]
    keep = sorted(set(positions) | set(j + 1 for j in positions))
    ref_all, dist_all, host = _resident_stream(engine, h, w, B, keep)
    dist_b, prev0 = dist_all.slice(1, B + 1), dist_all.frame(0)

    def exp(j):
        return j, check.expected(None, host[j + 1][1], host[j][1], True, (), motion="farneback", dct_mode="full")
    pool = ThreadPoolExecutor(4)
    fut = pool.map(exp, positions)
    params = engine.make_params(dct_mode=N.DCT_FULL, motion_mode=N.MOTION_FARNEBACK)
    c = engine.complexity(dist_b, prev0=prev0, mask=N.M_ALL, params=params)
    assert c.shape == (B,) and not c["hyst_overflow"].any() and (c["has_prev"] == 1).all()
    rel = np.abs(c["dct_energy"] - c["sum_gray2"].astype(np.float64)) / c["sum_gray2"].astype(np.float64)
    assert rel.max() < check.RTOL                                  # Parseval, every frame
    assert (c["temporal_dct_l1"] > 0).all() and (c["flow_mag_mean"] > 0).all() and (c["sad_blocks"] == 0).all()
    bad = {}
    for j, e in fut:
        notes = []
        m = check.compare(e, c[j], None, "gauss", notes=notes)
        if m or notes:
            bad[j] = (m, notes)
    pool.shutdown()
    assert not bad, bad


def test_farneback_chunk_seam_at_1080p(engine):
    """The Farneback pyramid works on chunks of at most 12 GiB of scratch (105 pairs of 1080p at 59 bytes per pixel and
    pair): a 125-pair batch is two chunks (105 + 20) with different strip geometries, partial-sum counts, a halo plane at
    the seam and - with VQA_OPT_OVERLAP - the second chunk's expansions waiting for the first chunk's iterations.  Oracle at
    both sides of the seam and at both ends; every pair positive."""
    from rtvqa_amd import _native as N
    h, w, B, positions = 1080, 1920, 125, [0, 104, 105, 124]
    keep = sorted(set(positions) | set(j + 1 for j in positions))
    ref_all, dist_all, host = _resident_stream(engine, h, w, B, keep)
    del ref_all
    dist_b, prev0 = dist_all.slice(1, B + 1), dist_all.frame(0)

    def exp(j):
        from oracle import c_oracle as co
        return j, co.farneback(co.bgr2gray(host[j][1]), co.bgr2gray(host[j + 1][1]))
    pool = ThreadPoolExecutor(4)
    fut = pool.map(exp, positions)
    c = engine.complexity(dist_b, prev0=prev0, mask=N.M_MOTION, motion_mode=N.MOTION_FARNEBACK)
    assert c.shape == (B,) and (c["flow_mag_mean"] > 0).all() and (c["has_prev"] == 1).all()
    for j, want in fut:
        got = float(c[j]["flow_mag_mean"])
        assert abs(got - want) <= check.RTOL * want, (j, got, want)
    pool.shutdown()


def test_farneback_2160p_against_the_oracle(engine):
    """The reference-true motion metric at 2160p (BASELINE configs[3]'s geometry): 16 column blocks per pair, five pyramid
    levels' worth of tile shapes in the level kernel; three pairs, oracle on the first and the last."""
    from rtvqa_amd import _native as N
    h, w, B, positions = 2160, 3840, 3, [0, 2]
    keep = sorted(set(positions) | set(j + 1 for j in positions))
    ref_all, dist_all, host = _resident_stream(engine, h, w, B, keep)
    del ref_all
    dist_b, prev0 = dist_all.slice(1, B + 1), dist_all.frame(0)

    def exp(j):
        from oracle import c_oracle as co
        return j, co.farneback(co.bgr2gray(host[j][1]), co.bgr2gray(host[j + 1][1]))
    pool = ThreadPoolExecutor(2)
    fut = pool.map(exp, positions)
    c = engine.complexity(dist_b, prev0=prev0, mask=N.M_MOTION, motion_mode=N.MOTION_FARNEBACK)
    for j, want in fut:
        got = float(c[j]["flow_mag_mean"])
        assert abs(got - want) <= check.RTOL * want, (j, got, want)
    pool.shutdown()
