"""Generate tests/golden/ref_pipeline.json: outputs of the REAL reference's own Python for everything it
computes itself, with the third-party cv2 calls served by this repository's restatement.

TEST INFRASTRUCTURE, build container only (/root/reference does not travel; the JSON does).

What this pins, and what it does not.  `import cv2` fails here (ModuleNotFoundError; opencv_python is not
installed and there is no network), so the reference's pixel arithmetic itself cannot run.  Everything AROUND
the cv2 calls is the reference's own code and can: frame selection and pairing (read_frame_pairs :76-111,
1-based `% k`), timestamp phase (extract_frame_timestamps :38-73, 0-based `% k`), the temporal-DCT priming
rule (calculate_temporal_dct :506-541), the order and dtype of the float tails (process_histogram_frame
:413-414, process_color_histogram_frame :467-473, process_dct_frame :363-364, process_edge_frame :504,
process_frame_complexity :342-343, process_temporal_dct_frame :574-579), the process-pool dispatcher, EWM
pooling and the tuple order (:246-310).  This script registers a FUNCTIONAL module object as sys.modules["cv2"]
whose functions delegate to oracle/c_oracle.py (VideoCapture over an ndarray clip; cvtColor / resize / dct /
Canny / calcHist / calcOpticalFlowFarneback / cartToPolar / ORB_create), imports the reference UNCHANGED and
runs those functions.  The fixtures therefore hold "the reference's orchestration and float tails applied to
the restated pixel kernels": orchestration and tails are pinned by the real reference; the pixel kernels
themselves (BGR2GRAY, resize, Canny, cv2.dct, FAST/ORB, Farneback) remain parity-unpinned.

Inputs are seeded synthetic clips (rtvqa_amd.synth, integer-only); the JSON stores generator arguments and a
SHA-256 of the bytes, not the frames.  No reference source text is copied.
Usage: python oracle/gen_golden_pipeline.py
"""
import hashlib
import json
import os
import sys
import tempfile
import types
import warnings

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(REPO, "tests", "golden", "ref_pipeline.json")
sys.path.insert(0, REPO)

from oracle import c_oracle as co  # noqa: E402
from rtvqa_amd import synth  # noqa: E402

CLIPS = {}  # path -> (frames [N,H,W,3] u8, fps)


def make_clip(spec):
    """spec: dict(kind, n, h, w, seed) -> uint8 [n,h,w,3].  Shared with tests/test_golden_pipeline.py."""
    if spec["kind"] == "natural":
        return synth.s_natural(spec["n"], spec["h"], spec["w"], seed=spec["seed"])
    if spec["kind"] == "noise":
        return synth.s_noise(spec["n"], spec["h"], spec["w"], seed=spec["seed"])
    if spec["kind"] == "natural+noise":  # textured content with grain: non-trivial Canny/FAST responses
        a = synth.s_natural(spec["n"], spec["h"], spec["w"], seed=spec["seed"]).astype(np.int16)
        b = synth.s_noise(spec["n"], spec["h"], spec["w"], seed=spec["seed"] + 1).astype(np.int16)
        return np.clip(a + (b >> 3) - 16, 0, 255).astype(np.uint8)
    raise KeyError(spec["kind"])


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


# ----------------------------------------------------------------------------------------------------------
# the functional cv2 stand-in (delegates to the restatement; asserts the argument values the reference passes)
# ----------------------------------------------------------------------------------------------------------
class _VideoCapture:
    def __init__(self, path):
        self._clip, self._fps = CLIPS.get(path, (None, 30.0))
        self._i = 0  # frames handed out so far

    def isOpened(self):
        return self._clip is not None

    def read(self):
        if self._clip is None or self._i >= len(self._clip):
            return False, None
        f = self._clip[self._i]
        self._i += 1
        return True, f

    def get(self, prop):
        assert prop == 0  # CAP_PROP_POS_MSEC: timestamp of the frame just read
        return float(self._i - 1) * 1000.0 / self._fps

    def release(self):
        self._clip = None


class _ORB:
    def detectAndCompute(self, gray, mask):
        assert mask is None and gray.shape == (64, 64) and gray.dtype == np.uint8
        n = co.orb64_count(gray)[0]
        return [object()] * n, None


def _cvtColor(img, code):
    assert code == 6 and img.ndim == 3 and img.shape[2] == 3 and img.dtype == np.uint8
    return co.bgr2gray(img)


def _resize(img, dsize):
    assert img.dtype == np.uint8
    return co.resize_linear(img, int(dsize[0]), int(dsize[1]))


def _dct(x):
    assert x.dtype == np.float32 and x.ndim == 2
    return co.dct2_full(x)


def _canny(gray, lo, hi):
    assert gray.dtype == np.uint8 and gray.ndim == 2
    edges = co.canny(gray, int(lo), int(hi), want_map=True)[3]
    return np.where(edges > 0, 255, 0).astype(np.uint8)


def _calcHist(images, channels, mask, hist_size, ranges):
    assert mask is None and hist_size == [256] and ranges == [0, 256] and len(images) == 1 and len(channels) == 1
    img = images[0]
    assert img.dtype == np.uint8
    if img.ndim == 2:
        assert channels == [0]
        counts = co.hist_u8(img)
    else:
        counts = co.hist_u8(img, offset=channels[0], step=img.shape[2])
    return counts.astype(np.float32).reshape(256, 1)  # calcHist returns a float32 column


def _farneback(prev, nxt, flow, pyr_scale, levels, winsize, iterations, poly_n, poly_sigma, flags):
    assert flow is None and (pyr_scale, levels, winsize, iterations, poly_n, poly_sigma, flags) == \
        (0.5, 3, 15, 3, 5, 1.2, 0)
    return co.farneback(prev, nxt, want_flow=True)[1]


def _cartToPolar(x, y):
    x, y = np.asarray(x, np.float32), np.asarray(y, np.float32)
    return np.sqrt(x * x + y * y), np.arctan2(y, x).astype(np.float32)


def install_cv2():
    cv2 = types.ModuleType("cv2")
    cv2.__doc__ = "functional stand-in serving the reference's cv2 calls from oracle/ (gen_golden_pipeline.py)"
    cv2.COLOR_BGR2GRAY = 6
    cv2.CAP_PROP_POS_MSEC = 0
    cv2.VideoCapture = _VideoCapture
    cv2.cvtColor = _cvtColor
    cv2.resize = _resize
    cv2.dct = _dct
    cv2.Canny = _canny
    cv2.calcHist = _calcHist
    cv2.calcOpticalFlowFarneback = _farneback
    cv2.cartToPolar = _cartToPolar
    cv2.ORB_create = _ORB
    sys.modules["cv2"] = cv2
    return cv2


def scalar(v):
    """-> JSON {type, value}: the Python/NumPy type the reference returned and its exact value."""
    t = type(v).__name__
    if isinstance(v, (np.floating, float)):
        f = float(v)
        return {"type": t, "value": None if np.isnan(f) else f}
    return {"type": t, "value": int(v)}


FRAME_SPECS = [
    dict(kind="natural", n=3, h=120, w=160, seed=11),
    dict(kind="natural+noise", n=3, h=96, w=128, seed=12),
    dict(kind="noise", n=2, h=72, w=88, seed=13),
    dict(kind="natural+noise", n=2, h=270, w=480, seed=14),
]
RESIZES = [(64, 64), None, (48, 40)]  # None = native (cv2.resize degenerates to a copy)

CLIP_CASES = [
    # (clip spec, frame_interval, resize (w,h) or None=native, fps, smoothing)
    (dict(kind="natural+noise", n=45, h=120, w=160, seed=21), 10, (64, 64), 30.0, 0.8),
    (dict(kind="natural+noise", n=6, h=120, w=160, seed=22), 1, None, 30.0, 0.8),
    (dict(kind="natural", n=15, h=120, w=160, seed=23), 10, (64, 64), 30.0, 0.8),   # one selected frame: no pair
    (dict(kind="natural", n=25, h=120, w=160, seed=24), 10, (64, 64), 25.0, 0.8),   # one pair, no temporal sample
    (dict(kind="natural+noise", n=31, h=96, w=128, seed=25), 3, (64, 48), 24.0, 0.5),
    (dict(kind="noise", n=9, h=72, w=88, seed=26), 2, None, 30.0, 0.8),
]


def main():
    if not os.path.isdir("/root/reference"):
        sys.exit("/root/reference not present: golden vectors can only be generated in the build container")
    co.build()
    work = tempfile.mkdtemp()
    os.chdir(work)
    install_cv2()
    sys.path.insert(0, "/root/reference")
    import complexity_metrics as ref  # noqa: E402  (the real, unmodified reference module)
    assert ref.use_gpu is False

    import pandas
    g = {"generator": "oracle/gen_golden_pipeline.py", "reference": "zaki699/Real-Time-Video-Quality-Analysis",
         "numpy": np.__version__, "pandas": pandas.__version__, "synth_version": synth.GENERATOR_VERSION,
         "note": "reference orchestration + float tails over the restated cv2 kernels; pixel kernels unpinned"}

    # ---- (a) the real per-frame callables on seeded frames
    frames_out = []
    for spec in FRAME_SPECS:
        clip = make_clip(spec)
        for fi in range(spec["n"]):
            frame = clip[fi]
            for rs in RESIZES:
                w, h = rs if rs is not None else (spec["w"], spec["h"])
                resized = co.resize_linear(frame, w, h)
                gray_of_resized = co.bgr2gray(resized)
                rec = {"spec": spec, "frame": fi, "sha256": sha(frame), "resize": [w, h],
                       "gray_counts": [int(v) for v in co.hist_u8(gray_of_resized)],
                       "bgr_counts": [[int(v) for v in co.hist_u8(resized, offset=c, step=3)] for c in range(3)],
                       "process_dct_frame": scalar(ref.process_dct_frame(frame, w, h)),
                       "process_histogram_frame": scalar(ref.process_histogram_frame(frame, w, h)),
                       "process_color_histogram_frame": scalar(ref.process_color_histogram_frame(frame, w, h)),
                       "process_edge_frame": scalar(ref.process_edge_frame(frame, w, h)),
                       "process_orb_frame_for_parallel": scalar(ref.process_orb_frame_for_parallel(frame))}
                if fi > 0:
                    prev = clip[fi - 1]
                    rec["process_frame_complexity"] = scalar(ref.process_frame_complexity((frame, prev)))
                    pg = co.resize_linear(co.bgr2gray(prev), w, h)
                    cg = co.resize_linear(co.bgr2gray(frame), w, h)
                    rec["process_temporal_dct_frame"] = scalar(ref.process_temporal_dct_frame(pg, cg, w, h))
                frames_out.append(rec)
    g["frames"] = frames_out
    g["process_frame_complexity_none"] = scalar(ref.process_frame_complexity((None, None)))

    # ---- (b) the real aggregator, frame selection, timestamps and temporal priming on whole clips
    clips_out = []
    real_smooth = ref.smooth_data
    for k, (spec, interval, rs, fps, alpha) in enumerate(CLIP_CASES):
        clip = make_clip(spec)
        path = "clip%d.mp4" % k
        CLIPS[path] = (clip, fps)
        w, h = rs if rs is not None else (spec["w"], spec["h"])
        series = []

        def tap(data, a=0.8, _s=series):
            _s.append([float(v) for v in data])
            return real_smooth(data, a)

        ref.smooth_data = tap
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            tup = ref.calculate_average_scene_complexity(path, w, h, frame_interval=interval,
                                                         smoothing_factor=alpha, num_workers=2, batch_size=4)
        ref.smooth_data = real_smooth
        names = ("motion", "dct", "hist", "edge", "orb", "color", "temporal", "framerate")
        assert len(series) == 8
        pairs = ref.read_frame_pairs(path, interval)
        # which source frames were paired: identify by content hash against the clip
        index_of = {sha(f): i for i, f in enumerate(clip)}
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            temporal_alone = ref.calculate_temporal_dct(path, w, h, interval, alpha)
        clips_out.append({
            "spec": spec, "sha256": sha(clip), "frame_interval": interval, "resize": [w, h], "fps": fps,
            "smoothing_factor": alpha,
            "tuple": [scalar(v) for v in tup],
            "series": dict(zip(names, series)),
            "pairs": [[index_of[sha(a)], index_of[sha(b)]] for a, b in pairs],
            "timestamps": [float(t) for t in ref.extract_frame_timestamps(path, interval)],
            "calculate_temporal_dct": scalar(temporal_alone),
        })
    # an unopenable video: every reader returns [] (:56-58, :95-97)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        tup = ref.calculate_average_scene_complexity("absent.mp4", 64, 64, num_workers=1)
    g["unopenable"] = {"tuple": [scalar(v) for v in tup], "pairs": ref.read_frame_pairs("absent.mp4"),
                       "timestamps": ref.extract_frame_timestamps("absent.mp4")}
    g["clips"] = clips_out

    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, "w") as f:
        json.dump(g, f, indent=None, separators=(",", ":"))
        f.write("\n")
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
