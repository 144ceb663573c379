"""Reference-shaped CPU pipeline built from the C oracle's primitives.

TEST INFRASTRUCTURE ONLY (see vqa_oracle.c).  Each function follows the call
order, argument values and dtype flow of the reference function it is named
after (citations per function) with cv2.* replaced by the oracle's restatement
of the same OpenCV routine.  Used by tests as the checker and by bench.py as the
"port" CPU baseline, driven by a dispatcher that reproduces process_in_batches'
process-pool behaviour (complexity_metrics.py:128-148).
"""
import functools
import multiprocessing
from concurrent.futures import ProcessPoolExecutor

import numpy as np

from . import c_oracle as co
from . import np_oracle as no


# complexity_metrics.py:346-364
def process_dct_frame(frame, resize_width, resize_height, dct_mode="full"):
    gray_frame = co.bgr2gray(frame)
    gray_frame = co.resize_linear(gray_frame, resize_width, resize_height)
    if dct_mode == "full":
        dct_frame = co.dct2_full(np.float32(gray_frame))  # :363
        return np.sum(dct_frame ** 2)                      # :364 — float32 pairwise sum, as NumPy does it
    return np.float32(co.dct8x8(None, gray_frame)[0])


# complexity_metrics.py:392-416
def process_histogram_frame(frame, resize_width, resize_height):
    frame_resized = co.resize_linear(frame, resize_width, resize_height)
    gray_frame = co.bgr2gray(frame_resized)
    return no.gray_entropy_from_counts(co.hist_u8(gray_frame))


# complexity_metrics.py:418-475
def process_color_histogram_frame(frame, resize_width, resize_height):
    resized_frame = co.resize_linear(frame, resize_width, resize_height)
    return no.color_entropy_from_counts([co.hist_u8(resized_frame, offset=c, step=3) for c in range(3)])


# complexity_metrics.py:477-504
def process_edge_frame(frame, resize_width, resize_height):
    frame_resized = co.resize_linear(frame, resize_width, resize_height)
    gray_frame = co.bgr2gray(frame_resized)
    return np.int64(co.canny(gray_frame, 100, 200)[0])


# complexity_metrics.py:367-389
def process_orb_frame_for_parallel(frame):
    gray_frame = co.bgr2gray(co.resize_linear(frame, 64, 64))
    return co.orb64_count(gray_frame)[0]


# complexity_metrics.py:313-343 — block-SAD substitute (spec: vqa_oracle.c vqo_block_sad), or with
# motion="farneback" the reference's own metric (vqo_farneback_mean_mag)
def process_frame_complexity(frame_pair, sad_range=7, motion="sad"):
    frame, prev_frame = frame_pair
    if frame is None or prev_frame is None:
        return 0.0
    curr_gray = co.bgr2gray(frame)
    prev_gray = co.bgr2gray(prev_frame)
    if motion == "farneback":
        flow = co.farneback(prev_gray, curr_gray, want_flow=True)[1]      # :340
        fx, fy = flow[..., 0], flow[..., 1]
        return np.mean(np.sqrt(fx * fx + fy * fy))                        # :342-343 (cartToPolar magnitude, f32)
    nb, _sad, hist = co.block_sad(prev_gray, curr_gray, sad_range)
    return np.float32(no.motion_mag_from_hist(hist, nb))


# complexity_metrics.py:543-579
def process_temporal_dct_frame(prev_gray_frame, curr_gray_frame, resize_width, resize_height, dct_mode="full"):
    prev_gray_frame = co.resize_linear(prev_gray_frame, resize_width, resize_height)
    curr_gray_frame = co.resize_linear(curr_gray_frame, resize_width, resize_height)
    if dct_mode == "full":
        prev_frame_dct = co.dct2_full(np.float32(prev_gray_frame))         # :574
        curr_frame_dct = co.dct2_full(np.float32(curr_gray_frame))         # :575
        return np.sum(np.abs(prev_frame_dct - curr_frame_dct))             # :578
    return np.float32(co.dct8x8(prev_gray_frame, curr_gray_frame)[1])


# complexity_metrics.py:128-148 — same executor, chunking and per-item pickling
def process_in_batches(frames, process_func, num_workers, batch_size=100, **kwargs):
    results = []
    with ProcessPoolExecutor(max_workers=num_workers) as executor:
        for i in range(0, len(frames), batch_size):
            batch = frames[i:i + batch_size]
            results.extend(executor.map(functools.partial(process_func, **kwargs), batch))
    return results


def serial_map(frames, process_func, num_workers=None, batch_size=100, **kwargs):
    return [process_func(f, **kwargs) for f in frames]


def selected_indices(num_frames, frame_interval):
    return np.arange(frame_interval - 1, num_frames, frame_interval)  # :103-104


# complexity_metrics.py:506-541
def calculate_temporal_dct(frames, resize_width, resize_height, frame_interval=10, smoothing_factor=0.8,
                           dct_mode="full"):
    idx = selected_indices(len(frames), frame_interval)
    prev_gray_frame = None
    energies = []
    for j in range(1, len(idx)):
        gray_frame = co.resize_linear(co.bgr2gray(frames[idx[j]]), resize_width, resize_height)
        if prev_gray_frame is not None:
            energies.append(process_temporal_dct_frame(prev_gray_frame, gray_frame, resize_width, resize_height,
                                                       dct_mode))
        prev_gray_frame = gray_frame
    sm = no.ewm_mean(energies, smoothing_factor)
    return (np.mean(sm) if len(sm) > 0 else 0.0), energies


# complexity_metrics.py:246-310 (fps from constant-rate timestamps)
def calculate_average_scene_complexity(frames, resize_width, resize_height, frame_interval=10, smoothing_factor=0.8,
                                       num_workers=None, batch_size=100, dct_mode="full", fps=30.0,
                                       dispatcher=serial_map, return_series=False, motion="sad"):
    idx = selected_indices(len(frames), frame_interval)
    frame_pairs = [(frames[idx[j]], frames[idx[j - 1]]) for j in range(1, len(idx))]
    if num_workers is None:
        num_workers = multiprocessing.cpu_count() // 2
    motion = dispatcher(frame_pairs, functools.partial(process_frame_complexity, motion=motion), num_workers, batch_size)
    sel = [p[0] for p in frame_pairs]
    kw = dict(resize_width=resize_width, resize_height=resize_height)
    dct = dispatcher(sel, functools.partial(process_dct_frame, dct_mode=dct_mode, **kw), num_workers, batch_size)
    hist = dispatcher(sel, functools.partial(process_histogram_frame, **kw), num_workers, batch_size)
    edge = dispatcher(sel, functools.partial(process_edge_frame, **kw), num_workers, batch_size)
    orb = dispatcher(sel, process_orb_frame_for_parallel, num_workers, batch_size)
    color = dispatcher(sel, functools.partial(process_color_histogram_frame, **kw), num_workers, batch_size)
    temporal, temporal_series = calculate_temporal_dct(frames, resize_width, resize_height, frame_interval,
                                                       smoothing_factor, dct_mode)
    ts = [float(i) * 1000.0 / fps for i in range(0, len(frames), frame_interval)]  # :65
    fpsv = [(1.0 / ((b - a) / 1000.0) if (b - a) > 0 else 0.0) for a, b in zip(ts[:-1], ts[1:])]

    def pooled(x):
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            return np.mean(no.ewm_mean(x, smoothing_factor))

    out = (pooled(motion), pooled(dct), pooled(hist), pooled(edge), pooled(orb), pooled(color), temporal,
           pooled(fpsv))
    if return_series:
        return out, dict(motion=motion, dct=dct, hist=hist, edge=edge, orb=orb, color=color, temporal=temporal_series)
    return out


# video_processing.py:270-297 — per-frame psnr/ssim numbers FFmpeg would log
def frame_quality(ref, dist, planes, ssim_mode="gauss"):
    """ref/dist: uint8 frame buffers (any shape); planes: (w,h,offset,row_stride,step) tuples.
    Returns (sse list, ssim list) per plane."""
    r = np.ascontiguousarray(ref).reshape(-1)
    d = np.ascontiguousarray(dist).reshape(-1)
    sse, ssim = [], []
    for (w, h, off, rs, step) in planes:
        a = np.lib.stride_tricks.as_strided(r[off:], shape=(h, w), strides=(rs, step), writeable=False)
        b = np.lib.stride_tricks.as_strided(d[off:], shape=(h, w), strides=(rs, step), writeable=False)
        sse.append(co.sse_plane(a, b))
        ssim.append(co.ssim_gauss(a, b) if ssim_mode == "gauss" else co.ssim_ffmpeg(a, b))
    return sse, ssim
