"""Independent NumPy/SciPy restatement of the same algorithms as vqa_oracle.c.

TEST INFRASTRUCTURE ONLY.  Exists so the C oracle is checked against a second,
differently-structured formulation (vectorised NMS + connected-component
labelling instead of a stack walk, scipy.fft instead of a cosine-matrix product,
candidate-major SAD instead of block-major, correlate1d instead of nested
loops).  Citations are to the reference call sites each function stands in for.
"""
import numpy as np
import scipy.fft
import scipy.ndimage


# cv2.cvtColor(BGR2GRAY) — complexity_metrics.py:327-328,358,405,493,530
def bgr2gray(bgr):
    b = bgr[..., 0].astype(np.int64)
    g = bgr[..., 1].astype(np.int64)
    r = bgr[..., 2].astype(np.int64)
    return ((b * 3735 + g * 19235 + r * 9798 + 16384) >> 15).astype(np.uint8)


def _tables(ssize, dsize, is_x):
    d = np.arange(dsize, dtype=np.float64)
    scale = 1.0 / (np.float64(dsize) / np.float64(ssize))
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int32)
    f = f - s.astype(np.float32)
    if is_x:
        lo = s < 0
        f[lo] = 0
        s[lo] = 0
        hi = s >= ssize - 1
        f[hi] = 0
        s[hi] = ssize - 1
    a0 = np.rint((np.float32(1.0) - f) * np.float32(2048)).astype(np.int32)
    a1 = np.rint(f * np.float32(2048)).astype(np.int32)
    return s, a0, a1


# cv2.resize default INTER_LINEAR — complexity_metrics.py:359,404,430,490,531
def resize_linear(img, dw, dh):
    sh, sw = img.shape[:2]
    if (sh, sw) == (dh, dw):
        return img.copy()
    x = img.reshape(sh, sw, -1).astype(np.int64)
    if sw == 2 * dw and sh == 2 * dh:
        out = (x[0::2, 0::2] + x[0::2, 1::2] + x[1::2, 0::2] + x[1::2, 1::2] + 2) >> 2
        return out.astype(np.uint8).reshape((dh, dw) + img.shape[2:])
    sx, a0, a1 = _tables(sw, dw, True)
    sy, b0, b1 = _tables(sh, dh, False)
    sx1 = np.minimum(sx + 1, sw - 1)
    rows = x[:, sx, :] * a0[None, :, None] + x[:, sx1, :] * a1[None, :, None]  # (sh, dw, cn)
    y0 = np.clip(sy, 0, sh - 1)
    y1 = np.clip(sy + 1, 0, sh - 1)
    s0 = rows[y0] >> 4
    s1 = rows[y1] >> 4
    out = (((b0[:, None, None] * s0) >> 16) + ((b1[:, None, None] * s1) >> 16) + 2) >> 2
    return out.astype(np.uint8).reshape((dh, dw) + img.shape[2:])


# cv2.dct(np.float32(x)) — complexity_metrics.py:363,574-575
def dct2_full(x):
    return scipy.fft.dctn(np.asarray(x, np.float64), norm="ortho")


def dct8x8_blocks(gray):
    h, w = gray.shape
    hp, wp = (h + 7) // 8 * 8, (w + 7) // 8 * 8
    p = np.zeros((hp, wp), np.float64)
    p[:h, :w] = gray
    b = p.reshape(hp // 8, 8, wp // 8, 8).transpose(0, 2, 1, 3)
    return scipy.fft.dctn(b, axes=(2, 3), norm="ortho")


def dct8x8(prev, curr):
    cc = dct8x8_blocks(curr)
    e = float(np.sum(cc * cc))
    l1 = float(np.sum(np.abs(dct8x8_blocks(prev) - cc))) if prev is not None else 0.0
    return e, l1


# cv2.calcHist — complexity_metrics.py:412,455-457
def hist_u8(a):
    return np.bincount(np.asarray(a, np.uint8).reshape(-1), minlength=256).astype(np.uint32)


# entropy tails, verbatim arithmetic of complexity_metrics.py:412-414 / :467-473 on float32 hists
def gray_entropy_from_counts(counts):
    hist = np.asarray(counts, np.float32).reshape(256, 1)
    hist = hist / hist.sum()
    return -np.sum(hist[hist > 0] * np.log2(hist[hist > 0]))


def color_entropy_from_counts(counts_bgr):
    hs = [np.asarray(c, np.float32).reshape(256, 1) for c in counts_bgr]
    sums = [h.sum() for h in hs]
    if sums[0] == 0 or sums[1] == 0 or sums[2] == 0:
        return float("nan")
    hs = [h / s for h, s in zip(hs, sums)]
    return -(np.sum(hs[0] * np.log2(hs[0] + 1e-8)) + np.sum(hs[1] * np.log2(hs[1] + 1e-8)) +
             np.sum(hs[2] * np.log2(hs[2] + 1e-8)))


# cv2.Canny(gray, 100, 200) — complexity_metrics.py:503-504
def canny(gray, low=100, high=200):
    g = np.pad(gray.astype(np.int32), 1, mode="edge")
    h, w = gray.shape
    def s(dy, dx):
        return g[1 + dy:1 + dy + h, 1 + dx:1 + dx + w]
    gx = (s(-1, 1) + 2 * s(0, 1) + s(1, 1)) - (s(-1, -1) + 2 * s(0, -1) + s(1, -1))
    gy = (s(1, -1) + 2 * s(1, 0) + s(1, 1)) - (s(-1, -1) + 2 * s(-1, 0) + s(-1, 1))
    mag = np.abs(gx) + np.abs(gy)
    mp = np.pad(mag, 1, mode="constant")
    def m(dy, dx):
        return mp[1 + dy:1 + dy + h, 1 + dx:1 + dx + w]
    ax = np.abs(gx).astype(np.int64)
    ay = np.abs(gy).astype(np.int64) << 15
    tg22 = ax * 13573
    tg67 = tg22 + (ax << 16)
    horiz = ay < tg22
    vert = (~horiz) & (ay > tg67)
    diag = ~(horiz | vert)
    opposite = (gx.astype(np.int16) ^ gy.astype(np.int16)) < 0
    keep_h = (mag > m(0, -1)) & (mag >= m(0, 1))
    keep_v = (mag > m(-1, 0)) & (mag >= m(1, 0))
    keep_d_same = (mag > m(-1, -1)) & (mag > m(1, 1))
    keep_d_opp = (mag > m(-1, 1)) & (mag > m(1, -1))
    keep = (mag > low) & ((horiz & keep_h) | (vert & keep_v) |
                          (diag & ~opposite & keep_d_same) | (diag & opposite & keep_d_opp))
    strong = keep & (mag > high)
    lab, n = scipy.ndimage.label(keep, structure=np.ones((3, 3), int))
    if n == 0:
        return 0, 0, 0, np.zeros((h, w), np.uint8)
    has_strong = np.zeros(n + 1, bool)
    has_strong[np.unique(lab[strong])] = True
    has_strong[0] = False
    edges = has_strong[lab]
    return int(edges.sum()), int(strong.sum()), int((keep & ~strong).sum()), (edges * 255).astype(np.uint8)


# block-SAD motion (substitute for complexity_metrics.py:313-343; spec in vqa_oracle.c)
def block_sad(prev, curr, rng=7):
    h, w = curr.shape
    nby, nbx = h // 16, w // 16
    if nby == 0 or nbx == 0:
        return 0, 0, np.zeros(129, np.uint32), np.zeros((0, 2), np.int8)
    c = curr[:nby * 16, :nbx * 16].astype(np.int32)
    P = prev.astype(np.int32)
    best = np.full((nby, nbx), np.iinfo(np.int64).max, np.int64)
    best_mv = np.zeros((nby, nbx, 2), np.int8)
    by = np.arange(nby)[:, None] * 16
    bx = np.arange(nbx)[None, :] * 16
    for dy in range(-rng, rng + 1):
        for dx in range(-rng, rng + 1):
            valid = (by + dy >= 0) & (by + 16 + dy <= h) & (bx + dx >= 0) & (bx + 16 + dx <= w)
            # shifted prev, out-of-frame samples never used by a valid block
            sh = np.zeros_like(c)
            ys = np.arange(nby * 16) + dy
            xs = np.arange(nbx * 16) + dx
            yv = (ys >= 0) & (ys < h)
            xv = (xs >= 0) & (xs < w)
            sh[np.ix_(yv, xv)] = P[np.ix_(ys[yv], xs[xv])]
            sad = np.abs(c - sh).reshape(nby, 16, nbx, 16).sum(axis=(1, 3)).astype(np.int64)
            d2 = dy * dy + dx * dx
            ridx = (dy + 8) * 16 + (dx + 8)
            key = (sad << 16) | (d2 << 8) | ridx
            key[~valid] = np.iinfo(np.int64).max
            upd = key < best
            best[upd] = key[upd]
            best_mv[upd] = (dy, dx)
    sad_sum = int((best >> 16).sum())
    d2 = (best_mv.astype(np.int32) ** 2).sum(axis=2)
    hist = np.bincount(d2.reshape(-1), minlength=129).astype(np.uint32)
    return nby * nbx, sad_sum, hist, best_mv.reshape(-1, 2)


def motion_mag_from_hist(d2_hist, nblocks):
    if nblocks == 0:
        return 0.0
    k = np.arange(129, dtype=np.float64)
    return float(np.sum(np.asarray(d2_hist, np.float64) * np.sqrt(k)) / nblocks)


# PSNR pieces (FFmpeg vf_psnr) / SSIM — video_processing.py:275-276
def sse_plane(a, b):
    d = a.astype(np.int64) - b.astype(np.int64)
    return int(np.sum(d * d))


def gauss11():
    k = np.arange(11, dtype=np.float64) - 5
    g = np.exp(-(k * k) / (2 * 1.5 * 1.5))
    return g / g.sum()


def ssim_gauss(a, b):
    g = gauss11()  # float64 taps: the definition (pinned by scikit-image, tests/golden/skimage_pins.json)
    x = a.astype(np.float64)
    y = b.astype(np.float64)
    def filt(z):
        z = scipy.ndimage.correlate1d(z, g, axis=0, mode="constant")
        z = scipy.ndimage.correlate1d(z, g, axis=1, mode="constant")
        return z[5:-5, 5:-5]
    mx, my = filt(x), filt(y)
    xx, yy, xy = filt(x * x), filt(y * y), filt(x * y)
    c1, c2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    vx, vy, cxy = xx - mx * mx, yy - my * my, xy - mx * my
    s = ((2 * mx * my + c1) * (2 * cxy + c2)) / ((mx * mx + my * my + c1) * (vx + vy + c2))
    return float(s.mean())


def ssim_ffmpeg(a, b):
    h, w = a.shape
    bh, bw = h >> 2, w >> 2
    x = a[:bh * 4, :bw * 4].astype(np.int64).reshape(bh, 4, bw, 4)
    y = b[:bh * 4, :bw * 4].astype(np.int64).reshape(bh, 4, bw, 4)
    s1 = x.sum(axis=(1, 3)); s2 = y.sum(axis=(1, 3))
    ss = (x * x).sum(axis=(1, 3)) + (y * y).sum(axis=(1, 3))
    s12 = (x * y).sum(axis=(1, 3))
    def pool(s):
        return s[:-1, :-1] + s[:-1, 1:] + s[1:, :-1] + s[1:, 1:]
    s1, s2, ss, s12 = pool(s1), pool(s2), pool(ss), pool(s12)
    c1 = int(.01 * .01 * 255 * 255 * 64 + .5)
    c2 = int(.03 * .03 * 255 * 255 * 64 * 63 + .5)
    vars_ = ss * 64 - s1 * s1 - s2 * s2
    covar = s12 * 64 - s1 * s2
    f = np.float32
    v = (f(2 * s1 * s2 + c1) * f(2 * covar + c2)) / (f(s1 * s1 + s2 * s2 + c1) * f(vars_ + c2))
    return float(v.astype(np.float64).mean())


# pandas .ewm(alpha, adjust=True).mean() — complexity_metrics.py:125
def ewm_mean(x, alpha=0.8):
    x = np.asarray(x, np.float64)
    out = np.empty_like(x)
    num = 0.0
    den = 0.0
    for i, v in enumerate(x):
        num = num * (1 - alpha) + v
        den = den * (1 - alpha) + 1.0
        out[i] = num / den
    return out


# FAST-9/16 in closed form (features2d/fast.cpp): score = max over the 16 runs of 9 contiguous circle
# pixels of min(v - ring) or min(ring - v), minus 1; corner iff that maximum exceeds the threshold.
_FAST_DX = (0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1)
_FAST_DY = (3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3)


def fast9_scores(gray, threshold=20):
    """score map (0 = not a corner) of every pixel at least 3 pixels inside the image."""
    g = np.asarray(gray, np.int32)
    h, w = g.shape
    sc = np.zeros((h, w), np.int32)
    if h < 7 or w < 7:
        return sc
    v = g[3:h - 3, 3:w - 3]
    ring = np.stack([g[3 + dy:h - 3 + dy, 3 + dx:w - 3 + dx] for dx, dy in zip(_FAST_DX, _FAST_DY)])
    d = v[None] - ring
    best = np.full(v.shape, -256, np.int32)
    for s0 in range(16):
        idx = [(s0 + j) % 16 for j in range(9)]
        best = np.maximum(best, np.maximum(d[idx].min(0), (-d[idx]).min(0)))
    sc[3:h - 3, 3:w - 3] = np.where(best > threshold, best - 1, 0)
    return sc


def fast9_nms(scores):
    """strict 3x3 non-max suppression of a FAST score map -> boolean keep map."""
    s = np.asarray(scores, np.int32)
    p = np.pad(s, 1)
    keep = s > 0
    h, w = s.shape
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            if dx or dy:
                keep &= s > p[1 + dy:1 + dy + h, 1 + dx:1 + dx + w]
    return keep


def orb64_count(gray64, threshold=20):
    """ORB keypoint count of a 64x64 image (see oracle/vqa_oracle.c vqo_orb64_count)."""
    keep = fast9_nms(fast9_scores(gray64, threshold))
    return int(keep[31:33, 31:33].sum())


# ---------------------------------------------------------------------------
# Farneback dense flow, vectorised in float64 with scipy.ndimage: an independent
# restatement of oracle/vqa_oracle.c's vqo_farneback_mean_mag (same published
# algorithm, different code shape, no float32 rounding) for cross-checking.  Valid for
# frames of at least 10x10 pixels: below that OpenCV's unsigned border test
# `(unsigned)(x - 5) >= (unsigned)(w - 10)` wraps and scales only some border pixels,
# which the C restatement (and the GPU) reproduce literally and this one does not.
# ---------------------------------------------------------------------------
def _fb_resize(img, dh, dw):
    img = np.asarray(img, np.float64)
    sh, sw = img.shape[:2]
    if (sh, sw) == (dh, dw):
        return img.copy()
    if sh == 2 * dh and sw == 2 * dw:
        return (img[0::2, 0::2] + img[0::2, 1::2] + img[1::2, 0::2] + img[1::2, 1::2]) * 0.25

    def axis(ssize, dsize, is_x):
        f = ((np.arange(dsize) + 0.5) * (1.0 / (dsize / ssize)) - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        f = (f - s).astype(np.float64)
        if is_x:
            f[s < 0] = 0
            s[s < 0] = 0
            f[s >= ssize - 1] = 0
            s[s >= ssize - 1] = ssize - 1
        return s, f
    xs, xf = axis(sw, dw, True)
    ys, yf = axis(sh, dh, False)
    x0, x1 = xs, np.minimum(xs + 1, sw - 1)
    y0, y1 = np.clip(ys, 0, sh - 1), np.clip(ys + 1, 0, sh - 1)
    sh_ = (1, dw) + (1,) * (img.ndim - 2)
    xf_, yf_ = xf.reshape(sh_), yf.reshape((dh, 1) + (1,) * (img.ndim - 2))
    r0 = img[y0][:, x0] * (1 - xf_) + img[y0][:, x1] * xf_
    r1 = img[y1][:, x0] * (1 - xf_) + img[y1][:, x1] * xf_
    return r0 * (1 - yf_) + r1 * yf_


def farneback_mean_mag(prev, curr, want_flow=False):
    import scipy.ndimage as ndi
    prev, curr = np.asarray(prev, np.float64), np.asarray(curr, np.float64)
    h, w = curr.shape
    levels, scale = 0, 1.0
    for k in range(3):
        scale *= 0.5
        if w * scale < 32 or h * scale < 32:
            break
        levels = k + 1
    n = 5
    x = np.arange(-n, n + 1, dtype=np.float64)
    g = np.exp(-x * x / (2 * 1.2 * 1.2)).astype(np.float32).astype(np.float64)
    g = (g / g.sum()).astype(np.float32).astype(np.float64)
    xg, xxg = (x * g).astype(np.float32).astype(np.float64), (x * x * g).astype(np.float32).astype(np.float64)
    G = np.zeros((6, 6))
    gg = np.outer(g, g)
    X, Y = np.meshgrid(x, x)
    G[0, 0], G[1, 1], G[3, 3], G[5, 5] = gg.sum(), (gg * X * X).sum(), (gg * X ** 4).sum(), (gg * X * X * Y * Y).sum()
    G[2, 2] = G[0, 3] = G[0, 4] = G[3, 0] = G[4, 0] = G[1, 1]
    G[4, 4] = G[3, 3]
    G[3, 4] = G[4, 3] = G[5, 5]
    iG = np.linalg.inv(G)
    ig11, ig03, ig33, ig55 = iG[1, 1], iG[0, 3], iG[3, 3], iG[5, 5]

    def polyexp(I):
        c = lambda a, k, ax: ndi.correlate1d(a, k, axis=ax, mode="nearest")
        r0, r1, r2 = c(I, g, 0), c(I, xg, 0), c(I, xxg, 0)
        b1, b2, b3 = c(r0, g, 1), c(r0, xg, 1), c(r1, g, 1)
        b4, b5, b6 = c(r0, xxg, 1), c(r2, g, 1), c(r1, xg, 1)
        return np.stack([b3 * ig11, b2 * ig11, b1 * ig03 + b5 * ig33, b1 * ig03 + b4 * ig33, b6 * ig55], -1)

    border = np.array([0.14, 0.14, 0.4472, 0.4472, 0.4472], np.float32).astype(np.float64)

    def update(R0, R1, flow):
        hh, ww = flow.shape[:2]
        yy, xx = np.mgrid[0:hh, 0:ww]
        dx, dy = flow[..., 0], flow[..., 1]
        fx, fy = (xx + dx).astype(np.float32).astype(np.float64), (yy + dy).astype(np.float32).astype(np.float64)
        x1, y1 = np.floor(fx).astype(np.int64), np.floor(fy).astype(np.int64)
        fx, fy = fx - x1, fy - y1
        ok = (x1 >= 0) & (x1 < ww - 1) & (y1 >= 0) & (y1 < hh - 1)
        xc, yc = np.clip(x1, 0, ww - 2), np.clip(y1, 0, hh - 2)
        a00, a01, a10, a11 = ((1 - fx) * (1 - fy))[..., None], (fx * (1 - fy))[..., None], ((1 - fx) * fy)[..., None], (fx * fy)[..., None]
        rr = a00 * R1[yc, xc] + a01 * R1[yc, xc + 1] + a10 * R1[yc + 1, xc] + a11 * R1[yc + 1, xc + 1]
        o = ok
        r2 = np.where(o, rr[..., 0], 0.0)
        r3 = np.where(o, rr[..., 1], 0.0)
        r4 = np.where(o, (R0[..., 2] + rr[..., 2]) * 0.5, R0[..., 2])
        r5 = np.where(o, (R0[..., 3] + rr[..., 3]) * 0.5, R0[..., 3])
        r6 = np.where(o, (R0[..., 4] + rr[..., 4]) * 0.25, R0[..., 4] * 0.5)
        r2 = (R0[..., 0] - r2) * 0.5
        r3 = (R0[..., 1] - r3) * 0.5
        r2 = r2 + r4 * dy + r6 * dx
        r3 = r3 + r6 * dy + r5 * dx
        sx, sy = np.ones(ww), np.ones(hh)
        for i in range(min(5, ww)):
            sx[i] *= border[i]
            sx[ww - 1 - i] *= border[i]
        for i in range(min(5, hh)):
            sy[i] *= border[i]
            sy[hh - 1 - i] *= border[i]
        sc = np.outer(sy, sx)
        r2, r3, r4, r5, r6 = r2 * sc, r3 * sc, r4 * sc, r5 * sc, r6 * sc
        return np.stack([r4 * r4 + r6 * r6, (r4 + r5) * r6, r5 * r5 + r6 * r6, r4 * r2 + r6 * r3, r6 * r2 + r5 * r3], -1)

    def blur_solve(M):
        ones = np.ones(15)
        v = ndi.correlate1d(ndi.correlate1d(M, ones, axis=0, mode="nearest"), ones, axis=1, mode="nearest") / 225.0
        idet = 1.0 / (v[..., 0] * v[..., 2] - v[..., 1] ** 2 + 1e-3)
        return np.stack([(v[..., 0] * v[..., 4] - v[..., 1] * v[..., 3]) * idet,
                         (v[..., 2] * v[..., 3] - v[..., 1] * v[..., 4]) * idet], -1)

    flow = None
    for k in range(levels, -1, -1):
        scale = 0.5 ** k
        sigma = (1.0 / scale - 1) * 0.5
        ks = max(int(np.rint(sigma * 5)) | 1, 3)
        lw, lh = int(np.rint(w * scale)), int(np.rint(h * scale))
        if sigma <= 0:
            kern = np.array([0.25, 0.5, 0.25])
        else:
            t = np.arange(ks) - ks // 2
            kern = np.exp(-0.5 * t * t / (sigma * sigma))
            kern = (kern / kern.sum()).astype(np.float32).astype(np.float64)
        flow = np.zeros((lh, lw, 2)) if flow is None else _fb_resize(flow, lh, lw) * 2.0
        R = []
        for img in (prev, curr):
            b = ndi.correlate1d(ndi.correlate1d(img, kern, axis=1, mode="mirror"), kern, axis=0, mode="mirror")
            R.append(polyexp(_fb_resize(b, lh, lw)))
        M = update(R[0], R[1], flow)
        for i in range(3):
            flow = blur_solve(M)
            if i < 2:
                M = update(R[0], R[1], flow)
    mag = np.sqrt(flow[..., 0] ** 2 + flow[..., 1] ** 2)
    return (float(mag.mean()), flow) if want_flow else float(mag.mean())
