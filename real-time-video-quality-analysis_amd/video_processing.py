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
import os
import re
import threading

import numpy as np

from . import _native as N
from . import stream
from .complexity_metrics import _open_frames
from .engine import DeviceFrames, bgr_planes, gray_planes, yuv420p_planes

LAYOUTS = {
    # name: (plane builder, component letters as FFmpeg prints them)
    "bgr24": (bgr_planes, "bgr"),     # packed BGR24 frames [N,H,W,3]; FFmpeg labels RGB components r,g,b
    "gray": (gray_planes, "y"),       # [N,H,W]
    "yuv420p": (yuv420p_planes, "yuv"),  # [N, H*W*3/2] planar
}
_SSIM_MODES = {"gauss": N.SSIM_GAUSS, "ffmpeg": N.SSIM_FFMPEG}


def _geometry(reference, layout, height, width):
    if isinstance(reference, DeviceFrames):
        return reference.h, reference.w
    if layout == "yuv420p":
        return height, width
    return reference.shape[1], reference.shape[2]


def _host_stream(a):
    if isinstance(a, DeviceFrames):
        return a
    if type(a).__module__.startswith("torch") and hasattr(a, "is_cuda"):
        from .complexity_metrics import _from_torch
        a = _from_torch(a)
        if isinstance(a, DeviceFrames):
            return a
    a = np.asarray(a)
    if a.dtype != np.uint8:  # a silent cast would turn float frames in 0..1 into all-zero planes
        raise ValueError("frames must be uint8 (got %s)" % a.dtype)
    return a


def frame_quality(reference, distorted, layout="bgr24", ssim_mode="gauss", height=None, width=None, engine=None,
                  batch_size=64, on_chunk=None, device=None):
    """Per-frame SSE and SSIM per plane.  Returns (sse [n,p] uint64, ssim [n,p] float64, plane sizes).
    One pass (stream.run): chunks of up to batch_size frame pairs alternate between two engines; host streams travel
    from the caller's pinned memory or through the pinned ring.  on_chunk(first_frame, sse, ssim) sees every finished
    chunk in frame order while the next one is on the GPU."""
    reference, distorted = _host_stream(reference), _host_stream(distorted)
    if not isinstance(reference, DeviceFrames) and reference.shape != distorted.shape:
        raise ValueError("ref and dist must have the same shape")
    h, w = _geometry(reference, layout, height, width)
    planes = LAYOUTS[layout][0](h, w)
    q, _ = stream.run(distorted, reference, quality=stream.Quality(planes, _SSIM_MODES[ssim_mode]),
                      batch_size=batch_size, engine=engine, on_quality=on_chunk, device=device)
    return q[0], q[1], [(p[0], p[1]) for p in planes]


# ---------------------------------------------------------------------------
# FFmpeg-format stats lines, a chunk of frames at a time
# ---------------------------------------------------------------------------
def _db(x):
    """10 log10(x) with inf for x = inf (FFmpeg prints "inf")"""
    with np.errstate(divide="ignore"):
        return 10.0 * np.log10(x)


def _psnr(mse, peak=255.0):
    """FFmpeg vf_psnr.c get_psnr(): 10*log10(max^2 / mse); mse == 0 -> inf (the expression the stats lines use)"""
    with np.errstate(divide="ignore"):
        return float(_db(np.float64(peak * peak) / np.float64(mse)))


def psnr_stats_lines(n0, sse, sizes, comps):
    """Lines n0.. of FFmpeg's psnr stats_file (vf_psnr.c) for sse [m,p]: per-component mse = sse/(w*h); mse_avg
    weights components by plane area; psnr = 10 log10(255^2 / mse) (get_psnr(); mse 0 -> inf); 2-decimal text."""
    sse = np.asarray(sse, np.float64).reshape(-1, len(sizes))
    areas = [w * h for w, h in sizes]
    total = float(sum(areas))
    mse = sse / np.asarray(areas, np.float64)
    mse_avg = np.zeros(len(sse))
    for j, a in enumerate(areas):
        mse_avg = mse_avg + mse[:, j] * (a / total)
    with np.errstate(divide="ignore"):
        cols = [np.arange(n0, n0 + len(sse), dtype=np.float64), mse_avg] + [mse[:, j] for j in range(len(areas))]
        cols += [_db(255.0 * 255.0 / mse_avg)] + [_db(255.0 * 255.0 / mse[:, j]) for j in range(len(areas))]
    fmt = ("n:%d mse_avg:%0.2f " + "".join("mse_%c:%%0.2f " % c for c in comps) + "psnr_avg:%0.2f "
           + "".join("psnr_%c:%%0.2f " % c for c in comps) + "\n")
    return (fmt * len(sse)) % tuple(np.column_stack(cols).ravel().tolist())


def ssim_stats_lines(n0, ssim, sizes, comps):
    """Lines n0.. of FFmpeg's ssim stats_file (vf_ssim.c): 'n:1 Y:0.99 U:.. V:.. All:0.99 (20.0)'."""
    ssim = np.asarray(ssim, np.float64).reshape(-1, len(sizes))
    areas = [w * h for w, h in sizes]
    total = float(sum(areas))
    allv = np.zeros(len(ssim))
    for j, a in enumerate(areas):
        allv = allv + ssim[:, j] * (a / total)
    with np.errstate(divide="ignore", invalid="ignore"):
        db = np.where(allv < 1.0, _db(1.0 / (1.0 - allv)), np.inf)
    cols = [np.arange(n0, n0 + len(ssim), dtype=np.float64)] + [ssim[:, j] for j in range(len(areas))] + [allv, db]
    fmt = "n:%d " + "".join("%c:%%f " % c.upper() for c in comps) + "All:%f (%f)\n"
    return (fmt * len(ssim)) % tuple(np.column_stack(cols).ravel().tolist())


def psnr_stats_line(n, sse_row, sizes, comps):
    return psnr_stats_lines(n, [list(sse_row)], sizes, comps)


def ssim_stats_line(n, ssim_row, sizes, comps):
    return ssim_stats_lines(n, [list(ssim_row)], sizes, comps)


class _StatsWriter:
    """Writes the two stats files chunk by chunk (stream.run's on_quality), in FFmpeg's component order."""

    def __init__(self, psnr_log, ssim_log, layout, sizes):
        comps = LAYOUTS[layout][1]
        # FFmpeg lists rgb components in r,g,b order whatever the packing
        self.order = [2, 1, 0] if layout == "bgr24" else list(range(len(comps)))
        self.names = "rgb" if layout == "bgr24" else comps
        self.sizes = [sizes[j] for j in self.order]
        self.fp, self.fs = open(psnr_log, "w"), open(ssim_log, "w")

    def __call__(self, first_frame, sse, ssim):
        self.fp.write(psnr_stats_lines(first_frame + 1, sse[:, self.order], self.sizes, self.names))
        self.fs.write(ssim_stats_lines(first_frame + 1, ssim[:, self.order], self.sizes, self.names))

    def close(self):
        self.fp.close()
        self.fs.close()


def _open_quality_stream(src, layout, height, width):
    """-> (frames, layout, height, width).  .y4m paths select the yuv420p layout by themselves; headerless .yuv (planar
    yuv420p) and .bgr / .bgr24 (packed) files take their geometry from height / width."""
    if isinstance(src, str) and src.endswith(".y4m"):
        from .frames import open_y4m
        arr, h, w, _fps = open_y4m(src)     # a memory map: the pass pages in what it gathers, nothing is read up front
        return arr, "yuv420p", h, w
    if isinstance(src, str) and src.endswith(".yuv"):
        if not height or not width:
            raise ValueError("a raw yuv420p stream needs height and width")
        from .frames import read_raw_yuv420p
        return read_raw_yuv420p(src, height, width), "yuv420p", height, width
    if layout == "bgr24":
        return _open_frames(src, height, width), layout, height, width
    if isinstance(src, str):
        raise ValueError("Unsupported file type. Please provide a .y4m / .yuv (yuv420p) or .npy / .bgr24 ([N,H,W,3] BGR) stream.")
    return src, layout, height, width


def run_ffmpeg_metrics(reference_video, distorted_video, psnr_log, ssim_log, vmaf_log, vmaf_model_path=None,
                       layout="bgr24", ssim_mode="gauss", height=None, width=None, batch_size=64, device=None):
    """video_processing.py:270-297 — PSNR and SSIM between two streams, one stats line per frame.
    Streams: [N,H,W,3] BGR arrays / .npy (components r,g,b as FFmpeg labels RGB input), planar yuv420p
    arrays with height/width, or .y4m files (components y,u,v — what FFmpeg sees for an H.264 clip)."""
    ref, layout, height, width = _open_quality_stream(reference_video, layout, height, width)
    dist, layout_d, _, _ = _open_quality_stream(distorted_video, layout, height, width)
    if layout_d != layout:
        raise ValueError("reference and distorted streams must share a pixel layout")
    h, w = _geometry(ref, layout, height, width)
    wr = _StatsWriter(psnr_log, ssim_log, layout, [(p[0], p[1]) for p in LAYOUTS[layout][0](h, w)])
    try:
        frame_quality(ref, dist, layout, ssim_mode, height, width, batch_size=batch_size, on_chunk=wr, device=device)
    finally:
        wr.close()
    return None


_csv_lock = threading.Lock()


def thread_safe_update_csv(metrics, csv_file="video_quality_data.csv"):
    """video_processing.py:44-68 — append one row, header only when the file is new (no pandas needed).  The "is the file
    new" test sits INSIDE the lock (the reference tests at :56 before taking it at :62, so two threads can both write a header)."""
    import csv
    with _csv_lock:
        exists = os.path.isfile(csv_file) and os.path.getsize(csv_file) > 0
        with open(csv_file, "a", newline="") as f:
            wr = csv.writer(f)
            if not exists:
                wr.writerow(list(metrics.keys()))
            wr.writerow(list(metrics.values()))


# the keys this build adds to the reference's config.json (config.json:1-7 keeps its five; SURVEY.md section 5 "Config / flag system")
MODE_KEYS = {
    # key: (allowed values, message in the reference's validate_config style)
    "ssim_mode": (("gauss", "ffmpeg"), "ssim_mode must be 'gauss' or 'ffmpeg'."),
    "pixfmt": ((None, "bgr24", "yuv420p", "gray"), "pixfmt must be 'bgr24', 'yuv420p' or 'gray'."),
    "dct_mode": ((None, "auto", "block8", "full"), "dct_mode must be 'auto', 'block8' or 'full'."),
    "motion": ((None, "sad", "farneback"), "motion must be 'sad' or 'farneback'."),
}


def _check_mode_keys(config):
    for key, (allowed, message) in MODE_KEYS.items():
        if key in config and config[key] not in allowed:
            raise ValueError(message)
    dev = config.get("device")
    if dev is not None and (isinstance(dev, bool) or not isinstance(dev, int) or dev < 0):
        raise ValueError("device must be a non-negative integer.")
    bs = config.get("batch_size", 100)
    if isinstance(bs, bool) or not isinstance(bs, int) or bs <= 0:
        raise ValueError("batch_size must be a positive integer.")
    for key in ("height", "width"):   # the geometry of headerless inputs (.yuv, .bgr24)
        v = config.get(key)
        if v is not None and (isinstance(v, bool) or not isinstance(v, int) or v <= 0):
            raise ValueError("height and width must be positive integers.")


def process_video_and_extract_metrics(input_video, encoded_video, config, csv_file="video_quality_data.csv",
                                      bitrate=0, frame_rate=30.0, column_order="reference", encoded_bgr=None,
                                      height=None, width=None):
    """video_processing.py:180-267 minus the libx264 encode and ffprobe steps (external codec, out of
    scope): both streams arrive decoded.  Quality metrics compare input vs encoded (:216); complexity is
    computed on the ENCODED stream (:242-247).  ONE pass (stream.run) serves both.

    input_video / encoded_video   the pair the quality filters compare:
        [N,H,W,3] BGR frames (.npy, array, torch tensor, DeviceFrames) - the complexity kernels then read the same
            encoded frames, every chunk is uploaded once; or
        the decoded planes FFmpeg's psnr / ssim filters see (:274-276): .y4m files or planar [N, H*W*3/2] yuv420p
            arrays (config pixfmt "yuv420p") - together with
    encoded_bgr                   the encoded stream as cv2.VideoCapture decodes it ([N,H,W,3] BGR: .npy, array, tensor,
        DeviceFrames; complexity_metrics.py:100), which the complexity half reads.  Per chunk the planar pair and the chunk's
        selected BGR frames are uploaded, each byte once.
    config   the reference's keys (crf, resize_width, resize_height, frame_interval, vmaf_model_path; config.json:1-7) plus
        batch_size, ssim_mode ("gauss" north_star's 11x11 Gaussian, default | "ffmpeg" vf_ssim's 8x8 integer windows),
        pixfmt (None: by input | "bgr24" | "yuv420p" | "gray"), dct_mode ("auto" default: full-frame up to 128x128, 8x8 blocks
        above | "block8" | "full" the reference's cv2.dct at any size), motion ("sad" north_star's block-SAD | "farneback" the
        reference's own; default: set_motion_mode / VQA_MOTION), device (GPU index; default VQA_DEVICE, LOCAL_RANK, 0) and
        height / width (the geometry of headerless inputs: raw .yuv planar pairs, raw .bgr24 streams).
    column_order="reference" keeps the reference's unpacking of the 8-tuple (:235-242), which shifts five labels
    (SURVEY.md §3.2); "fixed" uses the tuple's true order."""
    import tempfile
    import uuid
    from . import complexity_metrics as cm
    _check_mode_keys(config)
    crf = config.get("crf", 23)
    rw, rh = config.get("resize_width", 64), config.get("resize_height", 64)
    interval = config.get("frame_interval", 10)
    batch_size = config.get("batch_size", 100)
    ssim_mode = _SSIM_MODES[config.get("ssim_mode", "gauss")]
    dct_mode = cm._DCT_MODES[config.get("dct_mode")]
    motion_mode = cm.motion_mode_of(config.get("motion"))
    device = config.get("device")
    layout = config.get("pixfmt") or "bgr24"
    height, width = height or config.get("height"), width or config.get("width")   # (raw .yuv / .bgr24 files carry no header)
    uid = uuid.uuid4().hex
    tmp = tempfile.gettempdir()
    psnr_log, ssim_log, vmaf_log = (os.path.join(tmp, "%s_%s.log" % (k, uid)) for k in ("psnr", "ssim", "vmaf"))
    try:
        ref, layout, qh, qw = _open_quality_stream(input_video, layout, height, width)
        qenc, layout_d, _h, _w = _open_quality_stream(encoded_video, layout, qh, qw)
        if layout_d != layout:
            raise ValueError("reference and distorted streams must share a pixel layout")
        if layout == "bgr24" and encoded_bgr is None:
            enc, qdist = qenc, None       # (:216 and :242 read the same encoded stream: every chunk is uploaded once)
        else:
            if encoded_bgr is None:
                raise ValueError("a %s quality pair needs the encoded stream's BGR frames for the complexity half "
                                 "(complexity_metrics.py:100 reads cv2's BGR decode): pass encoded_bgr=" % layout)
            enc, qdist = _open_frames(encoded_bgr, qh or height, qw or width), _host_stream(qenc)
            ref = _host_stream(ref)
        eh, ew = (enc.h, enc.w) if isinstance(enc, DeviceFrames) else (enc.shape[1], enc.shape[2])
        if qdist is None:
            h, w = eh, ew
        else:
            h, w = _geometry(ref, layout, qh or eh, qw or ew)
            if layout == "yuv420p":
                from .frames import frame_bytes_yuv420p
                for a in (ref, qdist):
                    if not isinstance(a, DeviceFrames) and (a.ndim != 2 or a.shape[1] != frame_bytes_yuv420p(h, w)):
                        raise ValueError("yuv420p streams must be planar [N, H*W*3/2] uint8 arrays of the frames' geometry "
                                         "(%dx%d: %d bytes per frame)" % (w, h, frame_bytes_yuv420p(h, w)))
        planes = LAYOUTS[layout][0](h, w)
        wr = _StatsWriter(psnr_log, ssim_log, layout, [(p[0], p[1]) for p in planes])
        try:
            _q, series = stream.run(enc, ref, quality=stream.Quality(planes, ssim_mode),
                                    complexity=stream.Complexity((rw, rh), interval, dct_mode=dct_mode, motion_mode=motion_mode),
                                    batch_size=batch_size, on_quality=wr, qdist=qdist, device=device)
        finally:
            wr.close()
        resolution = "%dx%d" % (ew, eh)
        metrics = extract_metrics_from_logs(psnr_log, ssim_log, vmaf_log, input_video, crf, bitrate, resolution, frame_rate)
        t = cm.pool_series(series, enc, interval, batch_size=batch_size, fps=frame_rate)
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
    """video_processing.py:87-98 — same range checks, same messages; then this build's added keys in the same style."""
    if not (1 <= config.get("crf", 23) <= 51):
        raise ValueError("CRF value must be between 1 and 51.")
    if config.get("resize_width", 0) <= 0 or config.get("resize_height", 0) <= 0:
        raise ValueError("Resize dimensions must be positive integers.")
    if config.get("frame_interval", 10) <= 0:
        raise ValueError("Frame interval must be a positive integer.")
    if not isinstance(config.get("num_workers", (os.cpu_count() or 2) // 2), int):
        raise ValueError("num_workers must be an integer.")
    _check_mode_keys(config)  # this build's keys: ssim_mode, pixfmt, dct_mode, motion, device, batch_size


def main(argv=None):
    """video_processing.py:300-321 with decoded streams instead of a container + encode step:
        python -m rtvqa_amd.video_processing config.json input.npy encoded.npy [--csv out.csv]"""
    import argparse
    ap = argparse.ArgumentParser(description="Quality + complexity metrics of an (input, encoded) stream pair -> CSV row.")
    ap.add_argument("config_file")
    ap.add_argument("input_video", help="reference stream: .npy [N,H,W,3] BGR")
    ap.add_argument("encoded_video", help="distorted stream, same geometry (.npy BGR, or .y4m with --encoded-bgr)")
    ap.add_argument("--encoded-bgr", default=None, help="with a .y4m quality pair: the encoded stream's BGR frames (.npy) for the complexity half")
    ap.add_argument("--csv", default="video_quality_data.csv")
    ap.add_argument("--column-order", default="reference", choices=["reference", "fixed"])
    a = ap.parse_args(argv)
    m = process_video_and_extract_metrics(a.input_video, a.encoded_video, load_config(a.config_file), csv_file=a.csv,
                                          column_order=a.column_order, encoded_bgr=a.encoded_bgr)
    print(m)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
