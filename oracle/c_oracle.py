"""ctypes bindings for oracle/libvqa_oracle.so (the C restatement).

TEST INFRASTRUCTURE ONLY — see the header of vqa_oracle.c.  The product package
never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libvqa_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "vqa_oracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libvqa_oracle.so"])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        u8p = C.POINTER(C.c_uint8)
        L.vqo_bgr2gray.argtypes = [u8p, C.c_int, C.c_int, C.c_ssize_t, u8p, C.c_ssize_t]
        L.vqo_bgr2gray.restype = None
        L.vqo_resize_tables.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int16)]
        L.vqo_resize_tables.restype = None
        L.vqo_resize_linear.argtypes = [u8p, C.c_int, C.c_int, C.c_int, u8p, C.c_int, C.c_int]
        L.vqo_resize_linear.restype = C.c_int
        L.vqo_dct2_full.argtypes = [C.POINTER(C.c_float), C.c_int, C.c_int, C.POINTER(C.c_float)]
        L.vqo_dct2_full.restype = C.c_int
        L.vqo_dct_energy_full.argtypes = [u8p, C.c_int, C.c_int]
        L.vqo_dct_energy_full.restype = C.c_double
        L.vqo_temporal_dct_full.argtypes = [u8p, u8p, C.c_int, C.c_int]
        L.vqo_temporal_dct_full.restype = C.c_double
        L.vqo_dct8x8.argtypes = [u8p, u8p, C.c_int, C.c_int, C.c_ssize_t, C.POINTER(C.c_double)]
        L.vqo_dct8x8.restype = None
        L.vqo_hist_u8.argtypes = [u8p, C.c_size_t, C.c_int, C.POINTER(C.c_uint32)]
        L.vqo_hist_u8.restype = None
        L.vqo_canny_count.argtypes = [u8p, C.c_int, C.c_int, C.c_ssize_t, C.c_int, C.c_int, u8p,
                                      C.POINTER(C.c_long), C.POINTER(C.c_long)]
        L.vqo_canny_count.restype = C.c_long
        L.vqo_sobel_l1.argtypes = [u8p, C.c_int, C.c_int, C.c_ssize_t, C.POINTER(C.c_int16), C.POINTER(C.c_int16), C.POINTER(C.c_int32)]
        L.vqo_sobel_l1.restype = None
        L.vqo_block_sad.argtypes = [u8p, u8p, C.c_int, C.c_int, C.c_ssize_t, C.c_int,
                                    C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_int8)]
        L.vqo_block_sad.restype = C.c_int
        L.vqo_sse_plane.argtypes = [u8p, C.c_ssize_t, u8p, C.c_ssize_t, C.c_int, C.c_int, C.c_int]
        L.vqo_sse_plane.restype = C.c_uint64
        L.vqo_ssim_gauss.argtypes = [u8p, C.c_ssize_t, u8p, C.c_ssize_t, C.c_int, C.c_int, C.c_int]
        L.vqo_ssim_gauss.restype = C.c_double
        L.vqo_ssim_ffmpeg.argtypes = [u8p, C.c_ssize_t, u8p, C.c_ssize_t, C.c_int, C.c_int, C.c_int]
        L.vqo_ssim_ffmpeg.restype = C.c_double
        L.vqo_gauss11.argtypes = [C.POINTER(C.c_double)]
        L.vqo_gauss11.restype = None
        L.vqo_fast9.argtypes = [u8p, C.c_int, C.c_int, C.c_ssize_t, C.c_int, C.c_int, C.POINTER(C.c_int32), u8p]
        L.vqo_fast9.restype = C.c_long
        L.vqo_orb64_count.argtypes = [u8p, C.c_ssize_t, C.POINTER(C.c_int)]
        L.vqo_orb64_count.restype = C.c_int
        L.vqo_farneback_mean_mag.argtypes = [u8p, u8p, C.c_int, C.c_int, C.c_ssize_t, C.POINTER(C.c_float)]
        L.vqo_farneback_mean_mag.restype = C.c_double
        L.vqo_fb_prepare.argtypes = [C.c_int, C.c_double, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                     C.POINTER(C.c_float), C.POINTER(C.c_double)]
        L.vqo_fb_prepare.restype = None
        _lib = L
    return _lib


def _u8(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def _c(a, dtype=np.uint8):
    return np.ascontiguousarray(a, dtype=dtype)


def bgr2gray(bgr):
    bgr = _c(bgr)
    h, w, _ = bgr.shape
    out = np.empty((h, w), np.uint8)
    lib().vqo_bgr2gray(_u8(bgr), h, w, w * 3, _u8(out), w)
    return out


def resize_tables(ssize, dsize, is_x):
    ofs = np.empty(dsize, np.int32)
    coef = np.empty(2 * dsize, np.int16)
    lib().vqo_resize_tables(ssize, dsize, int(is_x), ofs.ctypes.data_as(C.POINTER(C.c_int32)),
                            coef.ctypes.data_as(C.POINTER(C.c_int16)))
    return ofs, coef.reshape(dsize, 2)


def resize_linear(img, dw, dh):
    """cv2.resize(img, (dw, dh)) — note OpenCV's (width, height) argument order."""
    img = _c(img)
    sh, sw = img.shape[:2]
    cn = 1 if img.ndim == 2 else img.shape[2]
    out = np.empty((dh, dw) if img.ndim == 2 else (dh, dw, cn), np.uint8)
    rc = lib().vqo_resize_linear(_u8(img), sh, sw, cn, _u8(out), dh, dw)
    if rc:
        raise RuntimeError("vqo_resize_linear failed: %d" % rc)
    return out


def dct2_full(x):
    x = _c(x, np.float32)
    h, w = x.shape
    y = np.empty_like(x)
    lib().vqo_dct2_full(x.ctypes.data_as(C.POINTER(C.c_float)), h, w, y.ctypes.data_as(C.POINTER(C.c_float)))
    return y


def dct_energy_full(gray):
    gray = _c(gray)
    return lib().vqo_dct_energy_full(_u8(gray), gray.shape[0], gray.shape[1])


def temporal_dct_full(prev, curr):
    prev, curr = _c(prev), _c(curr)
    return lib().vqo_temporal_dct_full(_u8(prev), _u8(curr), curr.shape[0], curr.shape[1])


def dct8x8(prev, curr):
    """-> (energy(curr), L1(prev,curr) in double, L1 with f32-rounded DCTs)."""
    curr = _c(curr)
    out = (C.c_double * 3)()
    if prev is None:
        lib().vqo_dct8x8(None, _u8(curr), curr.shape[0], curr.shape[1], curr.shape[1], out)
    else:
        prev = _c(prev)
        lib().vqo_dct8x8(_u8(prev), _u8(curr), curr.shape[0], curr.shape[1], curr.shape[1], out)
    return out[0], out[1], out[2]


def hist_u8(a, offset=0, step=1):
    a = _c(a).reshape(-1)
    n = (a.size - offset + step - 1) // step
    out = np.zeros(256, np.uint32)
    p = C.cast(C.c_void_p(a.ctypes.data + offset), C.POINTER(C.c_uint8))
    lib().vqo_hist_u8(p, n, step, out.ctypes.data_as(C.POINTER(C.c_uint32)))
    return out


def canny(gray, low=100, high=200, want_map=False):
    gray = _c(gray)
    h, w = gray.shape
    edges = np.empty((h, w), np.uint8) if want_map else None
    ns, nw = C.c_long(0), C.c_long(0)
    cnt = lib().vqo_canny_count(_u8(gray), h, w, w, low, high, _u8(edges) if want_map else None,
                                C.byref(ns), C.byref(nw))
    if cnt < 0:
        raise MemoryError
    return (cnt, ns.value, nw.value, edges) if want_map else (cnt, ns.value, nw.value)


def sobel_l1(gray):
    """cv2.Canny's gradient stage: (dx, dy) int16 Sobel 3x3 on a replicated border and |dx| + |dy| as int32."""
    gray = _c(gray)
    h, w = gray.shape
    dx, dy = np.zeros((h, w), np.int16), np.zeros((h, w), np.int16)
    mag = np.zeros((h, w), np.int32)
    lib().vqo_sobel_l1(_u8(gray), h, w, w, dx.ctypes.data_as(C.POINTER(C.c_int16)), dy.ctypes.data_as(C.POINTER(C.c_int16)),
                       mag.ctypes.data_as(C.POINTER(C.c_int32)))
    return dx, dy, mag


def block_sad(prev, curr, rng=7, want_mv=False):
    prev, curr = _c(prev), _c(curr)
    h, w = curr.shape
    sad = C.c_uint64(0)
    hist = np.zeros(129, np.uint32)
    nb = (h // 16) * (w // 16)
    mv = np.zeros((max(nb, 1), 2), np.int8)
    n = lib().vqo_block_sad(_u8(prev), _u8(curr), h, w, w, rng, C.byref(sad),
                            hist.ctypes.data_as(C.POINTER(C.c_uint32)),
                            mv.ctypes.data_as(C.POINTER(C.c_int8)) if want_mv else None)
    if n < 0:
        raise ValueError("bad range")
    return (n, sad.value, hist, mv[:nb]) if want_mv else (n, sad.value, hist)


def _plane_args(a, b):
    """a, b: 2-D uint8 views (possibly a channel slice of packed BGR)."""
    assert a.shape == b.shape and a.dtype == np.uint8 and b.dtype == np.uint8
    assert a.strides[1] == b.strides[1]
    h, w = a.shape
    pa = C.cast(C.c_void_p(a.ctypes.data), C.POINTER(C.c_uint8))
    pb = C.cast(C.c_void_p(b.ctypes.data), C.POINTER(C.c_uint8))
    return pa, a.strides[0], pb, b.strides[0], h, w, a.strides[1]


def sse_plane(a, b):
    return lib().vqo_sse_plane(*_plane_args(a, b))


def ssim_gauss(a, b):
    return lib().vqo_ssim_gauss(*_plane_args(a, b))


def ssim_ffmpeg(a, b):
    return lib().vqo_ssim_ffmpeg(*_plane_args(a, b))


def gauss11():
    g = (C.c_double * 11)()
    lib().vqo_gauss11(g)
    return np.array(g[:], np.float64)


def fast9(gray, threshold=20, nonmax=True):
    """cv2.FastFeatureDetector_create(threshold, nonmax).detect -> (n keypoints, score map, kept map)."""
    gray = _c(gray)
    h, w = gray.shape
    sc = np.zeros((h, w), np.int32)
    keep = np.zeros((h, w), np.uint8)
    n = lib().vqo_fast9(_u8(gray), h, w, w, int(threshold), int(bool(nonmax)),
                        sc.ctypes.data_as(C.POINTER(C.c_int32)), _u8(keep))
    if n < 0:
        raise MemoryError
    return n, sc, keep


def orb64_count(gray64):
    """len(cv2.ORB_create().detectAndCompute(gray64, None)[0]) on a 64x64 image -> (count, FAST response)."""
    gray64 = _c(gray64)
    assert gray64.shape == (64, 64)
    r = C.c_int(0)
    n = lib().vqo_orb64_count(_u8(gray64), 64, C.byref(r))
    if n < 0:
        raise MemoryError
    return n, r.value


def farneback(prev, curr, want_flow=False):
    """np.mean(|cv2.calcOpticalFlowFarneback(prev, curr, None, 0.5, 3, 15, 3, 5, 1.2, 0)|) (:340-343)."""
    prev, curr = _c(prev), _c(curr)
    h, w = curr.shape
    flow = np.zeros((h, w, 2), np.float32) if want_flow else None
    m = lib().vqo_farneback_mean_mag(_u8(prev), _u8(curr), h, w, w,
                                     flow.ctypes.data_as(C.POINTER(C.c_float)) if want_flow else None)
    if m < 0:
        raise MemoryError
    return (m, flow) if want_flow else m


def fb_prepare(n=5, sigma=1.2):
    g, xg, xxg = (np.zeros(2 * n + 1, np.float32) for _ in range(3))
    ig = np.zeros(4, np.float64)
    fp = C.POINTER(C.c_float)
    lib().vqo_fb_prepare(n, sigma, g.ctypes.data_as(fp), xg.ctypes.data_as(fp), xxg.ctypes.data_as(fp),
                         ig.ctypes.data_as(C.POINTER(C.c_double)))
    return g, xg, xxg, ig
