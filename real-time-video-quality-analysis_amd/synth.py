"""Seeded synthetic frame streams (integer-only, so every run and every rank
produces identical bytes).  These replace the decoded video the reference reads
with cv2.VideoCapture (complexity_metrics.py:76-111): decode is out of scope and
there is no codec in this image; benchmarks and tests use these streams.

  s_noise     uniform u8 noise — max-entropy bins, worst case for Canny fan-out
  s_natural   blurred multi-scale texture panned (2,1) px/frame, per-channel gain,
              four translating rectangles — smooth regions, real edges, real motion
  s_degenerate constant / checkerboard / step frames for the edge cases
  distort     deterministic +-3 grey-level perturbation (the 'encoded' stream)
"""
import numpy as np

GENERATOR_VERSION = 1


def s_noise(n, h, w, seed=1234, stream_id=0):
    rng = np.random.default_rng(seed + stream_id)
    return rng.integers(0, 256, size=(n, h, w, 3), dtype=np.uint8)


def _box3(a):
    """3x3 box blur on a uint32 array, wrap-around borders, integer arithmetic."""
    s = a + np.roll(a, 1, 0) + np.roll(a, -1, 0)
    s = s + np.roll(s, 1, 1) + np.roll(s, -1, 1)
    return s // 9


def _texture(h, w, seed):
    rng = np.random.default_rng(seed)
    acc = np.zeros((h, w), np.uint32)
    # octaves: coarse noise upsampled by pixel replication, then blurred
    for scale, weight in ((32, 8), (8, 4), (2, 2), (1, 1)):
        gh, gw = (h + scale - 1) // scale + 1, (w + scale - 1) // scale + 1
        g = rng.integers(0, 256, size=(gh, gw), dtype=np.uint32)
        up = np.repeat(np.repeat(g, scale, 0), scale, 1)[:h, :w]
        for _ in range(3):
            up = _box3(up)
        acc += up * weight
    lo, hi = int(acc.min()), int(acc.max())
    return ((acc - lo) * 255 // max(hi - lo, 1)).astype(np.uint8)


def s_natural(n, h, w, seed=1234, stream_id=0, t0=0):
    margin = 256
    tex = _texture(h + margin, w + margin, seed + 1000 * stream_id)
    out = np.empty((n, h, w, 3), np.uint8)
    rect_rng = np.random.default_rng(seed + 7 + stream_id)
    rects = []
    for _ in range(4):
        rh, rw = int(rect_rng.integers(h // 16 + 2, h // 5 + 3)), int(rect_rng.integers(w // 16 + 2, w // 5 + 3))
        rects.append((int(rect_rng.integers(0, h)), int(rect_rng.integers(0, w)), rh, rw,
                      int(rect_rng.integers(-3, 4)), int(rect_rng.integers(-4, 5)),
                      [int(v) for v in rect_rng.integers(0, 256, 3)]))
    for i in range(n):
        t = t0 + i
        oy, ox = (t * 1) % margin, (t * 2) % margin
        g = tex[oy:oy + h, ox:ox + w].astype(np.uint16)
        f = out[i]
        f[..., 0] = g
        f[..., 1] = (g * 7) >> 3
        f[..., 2] = (g * 3) >> 2
        for (y0, x0, rh, rw, vy, vx, col) in rects:
            y = (y0 + vy * t) % h
            x = (x0 + vx * t) % w
            f[y:min(y + rh, h), x:min(x + rw, w)] = col
    return out


def distort(ref, t0=0):
    """clip(ref + ((x*73856093 ^ y*19349663 ^ t*83492791) mod 7 - 3)) per pixel (same for B,G,R)."""
    n, h, w = ref.shape[:3]
    y = (np.arange(h, dtype=np.uint64) * np.uint64(19349663))[:, None]
    x = (np.arange(w, dtype=np.uint64) * np.uint64(73856093))[None, :]
    out = np.empty_like(ref)
    for i in range(n):
        t = np.uint64(t0 + i) * np.uint64(83492791)
        d = ((x ^ y ^ t) % np.uint64(7)).astype(np.int16) - 3
        out[i] = np.clip(ref[i].astype(np.int16) + (d[..., None] if ref.ndim == 4 else d), 0, 255).astype(np.uint8)
    return out


def s_degenerate(h, w):
    """dict name -> one BGR frame."""
    fr = {}
    fr["zeros"] = np.zeros((h, w, 3), np.uint8)
    fr["full"] = np.full((h, w, 3), 255, np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    cb = ((((yy >> 3) + (xx >> 3)) & 1) * 255).astype(np.uint8)
    fr["checker8"] = np.repeat(cb[..., None], 3, 2)
    st = np.where(xx < w // 2, 40, 200).astype(np.uint8)
    fr["vstep"] = np.repeat(st[..., None], 3, 2)
    ramp = (xx % 256).astype(np.uint8)
    fr["ramp"] = np.repeat(ramp[..., None], 3, 2)
    return fr
