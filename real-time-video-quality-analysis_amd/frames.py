"""Frame ingest without OpenCV / FFmpeg.

The reference reads frames with cv2.VideoCapture (complexity_metrics.py:76-111) and lets the ffmpeg
binary decode both inputs of the quality filters (video_processing.py:284-291).  Neither exists in the
target image and decode is out of scope, so streams arrive already decoded:

  .npy          [N,H,W,3] uint8 packed BGR24 (complexity path; what cv2 would have produced)
  .y4m          YUV4MPEG2, 8-bit C420* planar 4:2:0 (quality path: the planes FFmpeg's psnr/ssim see); open_y4m maps the
                file instead of reading it (a strided view: no host memory up front)
  .yuv / raw    headerless yuv420p with explicit width/height (mapped)
  .bgr / .bgr24 headerless packed BGR24 with explicit width/height (mapped)

Frames land in (optionally pinned) host buffers the engine can DMA from.
"""
import os
import re

import numpy as np


def frame_bytes_yuv420p(h, w):
    cw, ch = (w + 1) // 2, (h + 1) // 2
    return w * h + 2 * cw * ch


def _y4m_header(path):
    """-> (header length in bytes, height, width, fps)"""
    with open(path, "rb") as f:
        header = f.readline()
    if not header.startswith(b"YUV4MPEG2"):
        raise ValueError("not a YUV4MPEG2 stream: %s" % path)
    tok = header.decode("ascii", "replace").split()
    w = int(next(t[1:] for t in tok if t.startswith("W")))
    h = int(next(t[1:] for t in tok if t.startswith("H")))
    cs = next((t[1:] for t in tok if t.startswith("C")), "420jpeg")
    if not cs.startswith("420") or re.search(r"p1[0-6]", cs):
        raise ValueError("only 8-bit 4:2:0 Y4M is supported (got C%s)" % cs)
    fr = next((t[1:] for t in tok if t.startswith("F")), "30:1")
    num, den = (int(x) for x in fr.split(":"))
    return len(header), h, w, (num / den if den else 0.0)


def open_y4m(path, max_frames=None):
    """-> (frames [N, bytes_per_frame] uint8 in Y,U,V plane order, height, width, fps) WITHOUT reading the file: a strided view
    of a memory map (every frame sits a 6-byte FRAME line + bytes_per_frame after the previous one), so a clip of any length costs no
    host memory up front and the pass pages in what it gathers into the pinned ring (stream.py) - the reference hands the file
    to an ffmpeg subprocess that streams it the same way (video_processing.py:284-291).  Falls back to read_y4m (which parses
    frame by frame) when a frame header carries parameters, i.e. the frames are not equally spaced."""
    hl, h, w, fps = _y4m_header(path)
    fb = frame_bytes_yuv420p(h, w)
    size = os.path.getsize(path)
    n = (size - hl) // (fb + 6)
    if max_frames is not None:
        n = min(n, max_frames)
    if n <= 0:
        return np.zeros((0, fb), np.uint8), h, w, fps
    mm = np.memmap(path, dtype=np.uint8, mode="r")
    marks = np.lib.stride_tricks.as_strided(mm[hl:], shape=(n, 6), strides=(fb + 6, 1), writeable=False)
    if not (marks == np.frombuffer(b"FRAME\n", np.uint8)).all():
        del marks, mm
        return read_y4m(path, max_frames)
    frames = np.lib.stride_tricks.as_strided(mm[hl + 6:], shape=(n, fb), strides=(fb + 6, 1), writeable=False)
    return frames, h, w, fps


def read_y4m(path, max_frames=None, out=None):
    """-> (frames [N, bytes_per_frame] uint8 in Y,U,V plane order, height, width, fps), read into memory frame by frame
    (frame headers with parameters are accepted); `out`: a (pinned) array to read into."""
    hl, h, w, fps = _y4m_header(path)
    fb = frame_bytes_yuv420p(h, w)
    with open(path, "rb") as f:
        f.seek(hl)
        frames = []
        while max_frames is None or len(frames) < max_frames:
            line = f.readline()
            if not line:
                break
            if not line.startswith(b"FRAME"):
                raise ValueError("corrupt Y4M frame header")
            buf = f.read(fb)
            if len(buf) < fb:
                break
            frames.append(np.frombuffer(buf, np.uint8))
    arr = np.stack(frames) if frames else np.zeros((0, fb), np.uint8)
    if out is not None:
        out[:arr.shape[0]] = arr
        arr = out[:arr.shape[0]]
    return arr, h, w, fps


def write_y4m(path, frames, h, w, fps=(30, 1)):
    """frames: [N, bytes_per_frame] uint8 planar yuv420p."""
    fb = frame_bytes_yuv420p(h, w)
    frames = np.ascontiguousarray(frames, np.uint8).reshape(-1, fb)
    with open(path, "wb") as f:
        f.write(b"YUV4MPEG2 W%d H%d F%d:%d Ip A1:1 C420jpeg\n" % (w, h, fps[0], fps[1]))
        for fr in frames:
            f.write(b"FRAME\n")
            f.write(fr.tobytes())


def read_raw_yuv420p(path, h, w, max_frames=None, mmap=True):
    """headerless planar yuv420p -> [N, bytes_per_frame]; mapped, not read (mmap=False: loaded into memory)"""
    fb = frame_bytes_yuv420p(h, w)
    n = os.path.getsize(path) // fb
    if max_frames is not None:
        n = min(n, max_frames)
    if n <= 0:
        return np.zeros((0, fb), np.uint8)
    if mmap:
        return np.memmap(path, dtype=np.uint8, mode="r", shape=(n, fb))
    return np.fromfile(path, np.uint8, count=n * fb).reshape(n, fb)


def open_raw_bgr24(path, h, w, max_frames=None):
    """headerless packed BGR24 (`ffmpeg -f rawvideo -pix_fmt bgr24`, what cv2.VideoCapture.read yields frame by frame,
    complexity_metrics.py:100) -> [N,H,W,3] uint8, mapped, not read"""
    fb = int(h) * int(w) * 3
    if fb <= 0:
        raise ValueError("raw BGR24 streams need height and width")
    n = os.path.getsize(path) // fb
    if max_frames is not None:
        n = min(n, max_frames)
    if n <= 0:
        return np.zeros((0, int(h), int(w), 3), np.uint8)
    return np.memmap(path, dtype=np.uint8, mode="r", shape=(n, int(h), int(w), 3))


def bgr_to_yuv420p(bgr):
    """Deterministic integer BT.601 limited-range conversion used to derive synthetic yuv420p streams
    from synthetic BGR ones (NOT a restatement of any decoder): Y = (66R+129G+25B+128>>8)+16, chroma
    from the 2x2 mean."""
    bgr = np.asarray(bgr, np.uint8)
    n, h, w, _ = bgr.shape
    b, g, r = (bgr[..., c].astype(np.int32) for c in range(3))
    y = ((66 * r + 129 * g + 25 * b + 128) >> 8) + 16
    def sub(p):
        ph, pw = (h + 1) // 2 * 2, (w + 1) // 2 * 2
        q = np.pad(p, ((0, 0), (0, ph - h), (0, pw - w)), mode="edge")
        return (q[:, 0::2, 0::2] + q[:, 0::2, 1::2] + q[:, 1::2, 0::2] + q[:, 1::2, 1::2] + 2) >> 2
    rb, gb, bb = sub(r), sub(g), sub(b)
    u = ((-38 * rb - 74 * gb + 112 * bb + 128) >> 8) + 128
    v = ((112 * rb - 94 * gb - 18 * bb + 128) >> 8) + 128
    out = np.empty((n, frame_bytes_yuv420p(h, w)), np.uint8)
    out[:, :h * w] = np.clip(y, 0, 255).reshape(n, -1)
    cs = u.shape[1] * u.shape[2]
    out[:, h * w:h * w + cs] = np.clip(u, 0, 255).reshape(n, -1)
    out[:, h * w + cs:] = np.clip(v, 0, 255).reshape(n, -1)
    return out
