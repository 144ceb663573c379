"""Host-side mirror of the reference's quality-metric surface
(/root/reference/video_processing.py:145-177, :270-297), backed by the HIP engine.

run_ffmpeg_metrics keeps the reference's contract: it takes the reference and the
distorted stream plus the log paths, returns None and delivers its results as
FFmpeg-format stats files, so extract_metrics_from_logs' regular expressions
(:160, :166) parse them unchanged.  The streams are frame stacks (arrays / .npy)
instead of container files: decode is out of scope.  VMAF is out of scope: no
vmaf log is written, and — as in the reference when the file is absent (:169) —
the 'VMAF' key is simply missing.
"""
import math
import os
import re

import numpy as np

from . import _native as N
from .complexity_metrics import _open_frames, get_engine, get_engine_pair
from .engine import DeviceFrames, bgr_planes, gray_planes, yuv420p_planes

LAYOUTS = {
    # name: (plane builder, component letters as FFmpeg prints them)
    "bgr24": (bgr_planes, "bgr"),     # packed BGR24 frames [N,H,W,3]; FFmpeg labels RGB components r,g,b
    "gray": (gray_planes, "y"),       # [N,H,W]
    "yuv420p": (yuv420p_planes, "yuv"),  # [N, H*W*3/2] planar
}


def frame_quality(reference, distorted, layout="bgr24", ssim_mode="gauss", height=None, width=None, engine=None,
                  batch_size=64):
    """Per-frame SSE and SSIM per plane.  Returns (sse [n,p] uint64, ssim [n,p] float64, plane sizes)."""
    mode = {"gauss": N.SSIM_GAUSS, "ffmpeg": N.SSIM_FFMPEG}[ssim_mode]
    build, _ = LAYOUTS[layout]
    on_device = isinstance(reference, DeviceFrames)
    if on_device:
        h, w, n = reference.h, reference.w, reference.n
    else:
        reference = np.asarray(reference)
        distorted = np.asarray(distorted)
        n = reference.shape[0]
        if layout == "yuv420p":
            h, w = height, width
        else:
            h, w = reference.shape[1], reference.shape[2]
    planes = build(h, w)
    # host streams longer than one batch: two engines ping-pong so batch k+1's copy overlaps batch k's kernels
    engs = [engine or get_engine()] if (engine is not None or on_device or n <= batch_size) else list(get_engine_pair())
    sse, ssim, pending = [], [], []

    def collect(eng):
        res = eng.quality_wait()
        sse.append(res["sse"])
        ssim.append(res["ssim"])

    for k, a in enumerate(range(0, n, batch_size)):
        b = min(a + batch_size, n)
        eng = engs[k % len(engs)]
        if len(pending) == len(engs):
            collect(pending.pop(0))
        if on_device:
            eng.quality_submit(reference.slice(a, b), distorted.slice(a, b), planes, mode)
        else:
            eng.quality_submit(reference[a:b], distorted[a:b], planes, mode)
        pending.append(eng)
    while pending:
        collect(pending.pop(0))
    sizes = [(p[0], p[1]) for p in planes]
    return np.concatenate(sse), np.concatenate(ssim), sizes


def _psnr(mse, peak=255.0):
    # FFmpeg vf_psnr.c get_psnr(): 10*log10(max^2 / mse); mse == 0 -> inf
    return 10.0 * math.log10(peak * peak / mse) if mse > 0 else float("inf")


def psnr_stats_line(n, sse_row, sizes, comps):
    """One line of FFmpeg's psnr stats_file (vf_psnr.c): per-component mse = sse/(w*h);
    mse_avg weights components by plane area; 2-decimal text."""
    areas = [w * h for w, h in sizes]
    comp_mse = [float(s) / a for s, a in zip(sse_row, areas)]
    total = float(sum(areas))
    mse_avg = sum(m * (a / total) for m, a in zip(comp_mse, areas))
    parts = ["n:%d mse_avg:%0.2f " % (n, mse_avg)]
    parts += ["mse_%c:%0.2f " % (c, m) for c, m in zip(comps, comp_mse)]
    parts.append("psnr_avg:%0.2f " % _psnr(mse_avg))
    parts += ["psnr_%c:%0.2f " % (c, _psnr(m)) for c, m in zip(comps, comp_mse)]
    return "".join(parts) + "\n"


def ssim_stats_line(n, ssim_row, sizes, comps):
    """One line of FFmpeg's ssim stats_file (vf_ssim.c): 'n:1 Y:0.99 U:.. V:.. All:0.99 (20.0)'."""
    areas = [w * h for w, h in sizes]
    total = float(sum(areas))
    allv = sum(float(s) * (a / total) for s, a in zip(ssim_row, areas))
    db = 10.0 * math.log10(1.0 / (1.0 - allv)) if allv < 1.0 else float("inf")
    parts = ["n:%d " % n] + ["%c:%f " % (c.upper(), float(s)) for c, s in zip(comps, ssim_row)]
    parts.append("All:%f (%f)\n" % (allv, db))
    return "".join(parts)


def _open_quality_stream(src, layout, height, width):
    """-> (frames, layout, height, width).  .y4m paths select the yuv420p layout by themselves."""
    if isinstance(src, str) and src.endswith(".y4m"):
        from .frames import read_y4m
        arr, h, w, _fps = read_y4m(src)
        return arr, "yuv420p", h, w
    if layout == "bgr24":
        return _open_frames(src), layout, height, width
    return src, layout, height, width


def run_ffmpeg_metrics(reference_video, distorted_video, psnr_log, ssim_log, vmaf_log, vmaf_model_path=None,
                       layout="bgr24", ssim_mode="gauss", height=None, width=None):
    """video_processing.py:270-297 — PSNR and SSIM between two streams, one stats line per frame.
    Streams: [N,H,W,3] BGR arrays / .npy (components r,g,b as FFmpeg labels RGB input), planar yuv420p
    arrays with height/width, or .y4m files (components y,u,v — what FFmpeg sees for an H.264 clip)."""
    ref, layout, height, width = _open_quality_stream(reference_video, layout, height, width)
    dist, layout_d, _, _ = _open_quality_stream(distorted_video, layout, height, width)
    if layout_d != layout:
        raise ValueError("reference and distorted streams must share a pixel layout")
    sse, ssim, sizes = frame_quality(ref, dist, layout, ssim_mode, height, width)
    comps = LAYOUTS[layout][1]
    # FFmpeg lists rgb components in r,g,b order whatever the packing
    order = [2, 1, 0] if layout == "bgr24" else list(range(len(comps)))
    names = "rgb" if layout == "bgr24" else comps
    with open(psnr_log, "w") as f:
        for i in range(sse.shape[0]):
            f.write(psnr_stats_line(i + 1, [sse[i][j] for j in order], [sizes[j] for j in order], names))
    with open(ssim_log, "w") as f:
        for i in range(ssim.shape[0]):
            f.write(ssim_stats_line(i + 1, [ssim[i][j] for j in order], [sizes[j] for j in order], names))
    return None


def thread_safe_update_csv(metrics, csv_file="video_quality_data.csv"):
    """video_processing.py:44-68 — append one row, header only when the file is new (no pandas needed)."""
    import csv
    import threading
    global _csv_lock
    try:
        _csv_lock
    except NameError:
        _csv_lock = threading.Lock()
    exists = os.path.isfile(csv_file)
    with _csv_lock:
        with open(csv_file, "a", newline="") as f:
            wr = csv.writer(f)
            if not exists:
                wr.writerow(list(metrics.keys()))
            wr.writerow(list(metrics.values()))


def process_video_and_extract_metrics(input_video, encoded_video, config, csv_file="video_quality_data.csv",
                                      bitrate=0, frame_rate=30.0, column_order="reference"):
    """video_processing.py:180-267 minus the libx264 encode and ffprobe steps (external codec, out of
    scope): both streams arrive decoded.  Quality metrics compare input vs encoded (:216); complexity is
    computed on the ENCODED stream (:242-243).  column_order="reference" keeps the reference's unpacking of
    the 8-tuple (:235-242), which shifts five labels; "fixed" uses the tuple's true order."""
    import tempfile
    import uuid
    from . import complexity_metrics as cm
    crf = config.get("crf", 23)
    rw, rh = config.get("resize_width", 64), config.get("resize_height", 64)
    interval = config.get("frame_interval", 10)
    uid = uuid.uuid4().hex
    tmp = tempfile.gettempdir()
    psnr_log, ssim_log, vmaf_log = (os.path.join(tmp, "%s_%s.log" % (k, uid)) for k in ("psnr", "ssim", "vmaf"))
    try:
        run_ffmpeg_metrics(input_video, encoded_video, psnr_log, ssim_log, vmaf_log, config.get("vmaf_model_path"))
        enc = _open_frames(encoded_video)
        resolution = "%dx%d" % (enc.shape[2], enc.shape[1]) if hasattr(enc, "shape") else "%dx%d" % (enc.w, enc.h)
        metrics = extract_metrics_from_logs(psnr_log, ssim_log, vmaf_log, input_video, crf, bitrate, resolution, frame_rate)
        t = cm.calculate_average_scene_complexity(encoded_video, rw, rh, frame_interval=interval, fps=frame_rate)
        if column_order == "reference":   # (:235-242) motion, dct, temporal, hist, edge, orb, colour, fps
            names = ("Advanced Motion Complexity", "DCT Complexity", "Temporal DCT Complexity", "Histogram Complexity",
                     "Edge Detection Complexity", "ORB Feature Complexity", "Color Histogram Complexity",
                     "Framerate Variation")
        else:                             # (:301-310) the order the tuple really has
            names = ("Advanced Motion Complexity", "DCT Complexity", "Histogram Complexity", "Edge Detection Complexity",
                     "ORB Feature Complexity", "Color Histogram Complexity", "Temporal DCT Complexity",
                     "Framerate Variation")
        metrics.update(dict(zip(names, t)))
        thread_safe_update_csv(metrics, csv_file)
        return metrics
    finally:
        for p in (psnr_log, ssim_log, vmaf_log):
            if os.path.exists(p):
                os.remove(p)


def extract_metrics_from_logs(psnr_log, ssim_log, vmaf_log, video_file, crf, bitrate, resolution, frame_rate):
    """video_processing.py:145-177 — same keys, same regular expressions, first match only."""
    metrics = {"Bitrate (kbps)": bitrate, "Resolution (px)": resolution, "Frame Rate (fps)": frame_rate, "CRF": crf}
    if os.path.isfile(psnr_log):
        with open(psnr_log) as f:
            match = re.search(r"psnr_avg:(\s*\d+\.\d+)", f.read())
            if match:
                metrics["PSNR"] = float(match.group(1))
    if os.path.isfile(ssim_log):
        with open(ssim_log) as f:
            match = re.search(r"All:(\s*\d+\.\d+)", f.read())
            if match:
                metrics["SSIM"] = float(match.group(1))
    return metrics


def load_config(config_file):
    """video_processing.py:71-85 — JSON config, validated; same keys as the reference's config.json."""
    import json
    with open(config_file, "r") as f:
        config = json.load(f)
    validate_config(config)
    return config


def validate_config(config):
    """video_processing.py:87-98 — same range checks, same messages."""
    if not (1 <= config.get("crf", 23) <= 51):
        raise ValueError("CRF value must be between 1 and 51.")
    if config.get("resize_width", 0) <= 0 or config.get("resize_height", 0) <= 0:
        raise ValueError("Resize dimensions must be positive integers.")
    if config.get("frame_interval", 10) <= 0:
        raise ValueError("Frame interval must be a positive integer.")
    if not isinstance(config.get("num_workers", (os.cpu_count() or 2) // 2), int):
        raise ValueError("num_workers must be an integer.")


def main(argv=None):
    """video_processing.py:300-321 with decoded streams instead of a container + encode step:
        python -m rtvqa_amd.video_processing config.json input.npy encoded.npy [--csv out.csv]"""
    import argparse
    ap = argparse.ArgumentParser(description="Quality + complexity metrics of an (input, encoded) stream pair -> CSV row.")
    ap.add_argument("config_file")
    ap.add_argument("input_video", help="reference stream: .npy [N,H,W,3] BGR")
    ap.add_argument("encoded_video", help="distorted stream, same geometry")
    ap.add_argument("--csv", default="video_quality_data.csv")
    ap.add_argument("--column-order", default="reference", choices=["reference", "fixed"])
    a = ap.parse_args(argv)
    m = process_video_and_extract_metrics(a.input_video, a.encoded_video, load_config(a.config_file), csv_file=a.csv,
                                          column_order=a.column_order)
    print(m)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
