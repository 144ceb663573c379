"""Engine — thin Python owner of one vqa_ctx (one HIP device, one stream).

Host-side plumbing only: it hands pointers to the C ABI (include/vqa.h) and
turns the result records into NumPy arrays.  All pixel arithmetic happens in
the HIP kernels; nothing here computes a metric on the CPU.
"""
import ctypes as C

import numpy as np

from . import _native as N

# numpy mirror of vqa_frame_metrics (layout checked against ctypes at import)
FRAME_DTYPE = np.dtype([
    ("hist_gray", np.uint32, (256,)),
    ("hist_bgr", np.uint32, (3, 256)),
    ("sum_gray2", np.uint64),
    ("dct_energy", np.float64),
    ("temporal_dct_l1", np.float64),
    ("sad_sum", np.uint64),
    ("sad_blocks", np.uint32),
    ("mv_d2_hist", np.uint32, (129,)),
    ("edge_count", np.uint32),
    ("edge_strong", np.uint32),
    ("edge_weak", np.uint32),
    ("has_prev", np.uint32),
    ("hyst_steps", np.uint32), ("orb_keypoints", np.uint32), ("orb_response", np.uint32), ("hyst_overflow", np.uint32),
    ("flow_mag_mean", np.float64),
], align=True)
PLANE_DTYPE = np.dtype([("sse", np.uint64), ("ssim", np.float64)], align=True)
assert FRAME_DTYPE.itemsize == C.sizeof(N.VqaFrameMetrics), (FRAME_DTYPE.itemsize, C.sizeof(N.VqaFrameMetrics))
assert PLANE_DTYPE.itemsize == C.sizeof(N.VqaPlaneMetrics)


class DeviceFrames:
    """n packed BGR24 frames resident in device memory."""

    def __init__(self, ptr, n, h, w, frame_stride=None, row_stride=None, owner=None, channels=3):
        self.ptr, self.n, self.h, self.w = int(ptr), int(n), int(h), int(w)
        self.channels = channels
        self.row_stride = int(row_stride) if row_stride else self.w * channels
        self.frame_stride = int(frame_stride) if frame_stride else self.row_stride * self.h
        self._owner = owner  # keeps the allocation (torch tensor / DeviceBuffer) alive

    @classmethod
    def from_torch(cls, t):
        """Wrap a CUDA(HIP) uint8 torch tensor of shape [n,h,w,3] (or [n,h,w]) without copying."""
        assert t.is_cuda and t.dtype.__str__() == "torch.uint8" and t.is_contiguous()
        ch = t.shape[3] if t.dim() == 4 else 1
        return cls(t.data_ptr(), t.shape[0], t.shape[1], t.shape[2], owner=t, channels=ch)

    def frame(self, i):
        return DeviceFrames(self.ptr + i * self.frame_stride, 1, self.h, self.w, self.frame_stride, self.row_stride,
                            owner=self._owner, channels=self.channels)

    def slice(self, a, b):
        return DeviceFrames(self.ptr + a * self.frame_stride, b - a, self.h, self.w, self.frame_stride,
                            self.row_stride, owner=self._owner, channels=self.channels)

    def roi(self, y0, y1, x0, x1):
        """The window rows y0..y1, columns x0..x1 of every frame, in place (same memory, padded rows)."""
        assert 0 <= y0 < y1 <= self.h and 0 <= x0 < x1 <= self.w
        return DeviceFrames(self.ptr + y0 * self.row_stride + x0 * self.channels, self.n, y1 - y0, x1 - x0,
                            self.frame_stride, self.row_stride, owner=self._owner, channels=self.channels)


class DeviceBuffer:
    def __init__(self, engine, nbytes):
        self.engine, self.nbytes = engine, int(nbytes)
        p = C.c_void_p()
        N.check(engine.lib.vqa_alloc_device(engine.ctx, self.nbytes, C.byref(p)), "vqa_alloc_device", engine.ctx)
        self.ptr = p.value

    def free(self):
        if self.ptr:
            self.engine.lib.vqa_free_device(self.engine.ctx, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def bgr_planes(h, w):
    """B, G, R channels of packed BGR24 as three full-size planes."""
    return [(w, h, c, 3 * w, 3) for c in range(3)]


def gray_planes(h, w):
    return [(w, h, 0, w, 1)]


def yuv420p_planes(h, w):
    cw, ch = (w + 1) // 2, (h + 1) // 2
    return [(w, h, 0, w, 1), (cw, ch, w * h, cw, 1), (cw, ch, w * h + cw * ch, cw, 1)]


class Engine:
    def __init__(self, device=0):
        self.lib = N.load()
        ctx = C.c_void_p()
        st = self.lib.vqa_create(int(device), C.byref(ctx))
        if st != N.VQA_OK:
            raise N.VqaError(st, "vqa_create(device=%d)" % device)
        self.ctx = ctx
        self.device = int(device)

    def close(self):
        if getattr(self, "ctx", None):
            for p in getattr(self, "_pinned", []):
                self.lib.vqa_free_pinned(self.ctx, p)
            self._pinned = []
            self.lib.vqa_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- memory ----------------------------------------------------------
    def upload(self, arr):
        """Copy a host uint8 array [n,h,w,3] (or [n,h,w]) to device memory."""
        arr = np.ascontiguousarray(arr, dtype=np.uint8)
        buf = DeviceBuffer(self, arr.nbytes)
        N.check(self.lib.vqa_copy_h2d(self.ctx, buf.ptr, arr.ctypes.data, arr.nbytes), "vqa_copy_h2d", self.ctx)
        N.check(self.lib.vqa_sync(self.ctx), "vqa_sync", self.ctx)
        ch = arr.shape[3] if arr.ndim == 4 else 1
        return DeviceFrames(buf.ptr, arr.shape[0], arr.shape[1], arr.shape[2], owner=buf, channels=ch)

    def alloc_pinned(self, shape, dtype=np.uint8):
        """A NumPy array backed by page-locked host memory (hipHostMalloc): H2D from it is a true async DMA.
        The memory lives until the engine is closed."""
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = C.c_void_p()
        N.check(self.lib.vqa_alloc_pinned(self.ctx, max(nbytes, 1), C.byref(p)), "vqa_alloc_pinned", self.ctx)
        self._pinned = getattr(self, "_pinned", [])
        self._pinned.append(p.value)
        raw = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(max(nbytes, 1),))
        return raw[:nbytes].view(dtype).reshape(shape)

    def free_pinned(self, arr):
        """Give back an array alloc_pinned returned (it must not be used afterwards)."""
        p = arr.ctypes.data if hasattr(arr, "ctypes") else int(arr)
        if p in getattr(self, "_pinned", []):
            self._pinned.remove(p)
            N.check(self.lib.vqa_free_pinned(self.ctx, p), "vqa_free_pinned", self.ctx)

    def is_pinned(self, arr):
        """True if the array's memory - ALL of it, first byte to last - is page-locked and known to HIP (alloc_pinned, torch
        pin_memory, hipHostRegister): a submit / h2d from it is an asynchronous DMA; pageable memory, and an array that
        only starts inside a registered region, is staged through a pinned ring (stream.py)."""
        if getattr(arr, "nbytes", 0) == 0:
            return False
        if any(st < 0 for st in arr.strides):
            return False
        span = sum((n - 1) * st for n, st in zip(arr.shape, arr.strides)) + arr.itemsize  # first byte .. last byte of a strided view
        out = C.c_int(0)
        N.check(self.lib.vqa_host_is_pinned(self.ctx, arr.ctypes.data, span, C.byref(out)), "vqa_host_is_pinned", self.ctx)
        return bool(out.value)

    def h2d_async(self, dst_ptr, src_ptr, nbytes):
        """Enqueue a host-to-device copy on the engine's stream (a true DMA when the source is pinned)."""
        if nbytes:
            N.check(self.lib.vqa_copy_h2d(self.ctx, int(dst_ptr), int(src_ptr), int(nbytes)), "vqa_copy_h2d", self.ctx)

    def sync(self):
        N.check(self.lib.vqa_sync(self.ctx), "vqa_sync", self.ctx)

    def wait_for(self, other):
        """vqa_stream_wait: what is enqueued on this engine from now on starts after what `other` (an engine of the same device,
        e.g. the pass's copy lane) has enqueued so far - a device-side dependency, the host does not wait."""
        N.check(self.lib.vqa_stream_wait(self.ctx, other.ctx), "vqa_stream_wait", self.ctx)

    def drain(self):
        """Wait out whatever this engine still has pending (a quality and / or a complexity batch), discard the results and
        synchronise its streams: after a failure in the caller's loop nothing reads the caller's buffers any more and the
        engine is usable again.  Never raises."""
        for pend, wait in (("_pending_q", self.quality_wait), ("_pending_c", self.complexity_wait)):
            try:
                if getattr(self, pend, None):
                    wait()
            except Exception:
                setattr(self, pend, None)
        try:
            self.sync()
        except Exception:
            pass

    def trim(self):
        """vqa_trim: give back every scratch buffer, the result staging and every cached table of this (idle) engine.
        The next submit re-grows what it needs; results are unaffected."""
        N.check(self.lib.vqa_trim(self.ctx), "vqa_trim", self.ctx)

    # ---- options (include/vqa.h: none of them changes a result) --------------
    def set_option(self, option, value):
        N.check(self.lib.vqa_set_option(self.ctx, int(option), int(value)), "vqa_set_option", self.ctx)

    def get_option(self, option):
        v = C.c_int(0)
        N.check(self.lib.vqa_get_option(self.ctx, int(option), C.byref(v)), "vqa_get_option", self.ctx)
        return v.value

    def set_overlap(self, on):
        """Block-SAD and the Canny chain on side streams inside a complexity submit (default on)."""
        self.set_option(N.OPT_OVERLAP, 1 if on else 0)

    @property
    def stream(self):
        return self.lib.vqa_stream(self.ctx)

    # ---- complexity --------------------------------------------------------
    def make_params(self, resize=None, canny=(100, 200), sad_range=7, dct_mode=N.DCT_AUTO, motion_mode=N.MOTION_SAD):
        p = N.VqaParams()
        self.lib.vqa_default_params(C.byref(p))
        if resize:
            p.resize_w, p.resize_h = int(resize[0]), int(resize[1])
        p.canny_low, p.canny_high = int(canny[0]), int(canny[1])
        p.sad_range = int(sad_range)
        p.dct_mode = int(dct_mode)
        p.motion_mode = int(motion_mode)
        return p

    @staticmethod
    def _frames_args(frames, prev0):
        if isinstance(frames, DeviceFrames):
            if prev0 is not None and not isinstance(prev0, DeviceFrames):
                raise TypeError("prev0 must live where frames live (device)")
            pp = prev0.ptr if prev0 is not None else None
            return (frames.ptr, pp, N.VQA_MEM_DEVICE, frames.n, frames.h, frames.w, frames.frame_stride,
                    frames.row_stride, (frames, prev0))
        arr = np.asarray(frames)
        if arr.dtype != np.uint8:  # a silent cast would turn float frames in 0..1 into all-zero planes
            raise ValueError("frames must be uint8 (got %s): decoded 8-bit BGR frames, as cv2.VideoCapture.read yields" % arr.dtype)
        if arr.ndim == 3:
            arr = arr[None]
        if arr.ndim != 4 or arr.shape[3] != 3:
            raise ValueError("frames must be uint8 [n,h,w,3] packed BGR")
        # a strided selection of whole frames (clip[k-1::k]) and a region of interest (clip[:, y0:y1, x0:x1])
        # are passed as they are: frame_stride / row_stride do the stepping
        h_, w_ = arr.shape[1], arr.shape[2]
        st = arr.strides
        if not (st[2:] == (3, 1) and st[1] >= w_ * 3 and (arr.shape[0] == 1 or st[0] >= (h_ - 1) * st[1] + w_ * 3)):
            arr = np.ascontiguousarray(arr)
        rstride = arr.strides[1] if h_ > 1 else w_ * 3
        keep = [arr]
        pp = None
        if prev0 is not None:
            p0 = np.asarray(prev0)
            if p0.dtype != np.uint8:
                raise ValueError("prev0 must be uint8 (got %s)" % p0.dtype)
            if p0.shape != arr.shape[1:]:
                raise ValueError("prev0 must have the frames' geometry")
            if not (p0.strides[1:] == (3, 1) and (h_ == 1 or p0.strides[0] == rstride)):
                if rstride != w_ * 3:  # one row stride serves frames and prev0: fall back to compact copies
                    arr = np.ascontiguousarray(arr)
                    rstride = w_ * 3
                    keep[0] = arr
                p0 = np.ascontiguousarray(p0)
            keep.append(p0)
            pp = p0.ctypes.data
        n, h, w, _ = arr.shape
        fstride = arr.strides[0] if n > 1 else max(h * rstride, 1)
        return (arr.ctypes.data, pp, N.VQA_MEM_HOST, n, h, w, fstride, rstride, keep)

    def complexity_submit(self, frames, prev0=None, mask=N.M_ALL, params=None):
        fp, pp, kind, n, h, w, fs, rs, keep = self._frames_args(frames, prev0)
        params = params or self.make_params()
        st = self.lib.vqa_complexity_submit(self.ctx, fp, pp, kind, n, h, w, fs, rs, mask, C.byref(params))
        N.check(st, "vqa_complexity_submit", self.ctx)
        self._pending_c = (n, keep)
        return n

    def complexity_wait(self):
        n, _keep = self._pending_c
        out = np.zeros(n, dtype=FRAME_DTYPE)
        st = self.lib.vqa_complexity_wait(self.ctx, out.ctypes.data_as(C.POINTER(N.VqaFrameMetrics)), n)
        self._pending_c = None
        N.check(st, "vqa_complexity_wait", self.ctx)
        return out

    def complexity(self, frames, prev0=None, mask=N.M_ALL, params=None, **kw):
        """Run the selected complexity kernels over a batch; returns a structured array (FRAME_DTYPE)."""
        if params is None:
            params = self.make_params(**kw)
        self.complexity_submit(frames, prev0, mask, params)
        return self.complexity_wait()

    # ---- quality -----------------------------------------------------------
    def quality_submit(self, ref, dist, planes, ssim_mode=N.SSIM_GAUSS, frame_bytes=None):
        if isinstance(ref, DeviceFrames):
            assert isinstance(dist, DeviceFrames) and ref.n == dist.n
            rp, dp, kind, n = ref.ptr, dist.ptr, N.VQA_MEM_DEVICE, ref.n
            rfs, dfs = ref.frame_stride, dist.frame_stride
            keep = (ref, dist)
        else:
            ref = np.ascontiguousarray(ref, dtype=np.uint8)
            dist = np.ascontiguousarray(dist, dtype=np.uint8)
            if ref.shape != dist.shape:
                raise ValueError("ref and dist must have the same shape")
            n = ref.shape[0]
            rfs = dfs = frame_bytes or (ref.nbytes // n)
            rp, dp, kind = ref.ctypes.data, dist.ctypes.data, N.VQA_MEM_HOST
            keep = (ref, dist)
        descs = (N.VqaPlaneDesc * len(planes))()
        for i, (w, h, off, rs, step) in enumerate(planes):
            descs[i].width, descs[i].height, descs[i].offset = w, h, off
            descs[i].row_stride, descs[i].pixel_step = rs, step
        st = self.lib.vqa_quality_submit(self.ctx, rp, dp, kind, n, rfs, dfs, descs, len(planes), ssim_mode)
        N.check(st, "vqa_quality_submit", self.ctx)
        self._pending_q = (n, len(planes), keep)

    def quality_wait(self):
        n, npl, _keep = self._pending_q
        out = np.zeros(n * npl, dtype=PLANE_DTYPE)
        st = self.lib.vqa_quality_wait(self.ctx, out.ctypes.data_as(C.POINTER(N.VqaPlaneMetrics)), n * npl)
        self._pending_q = None
        N.check(st, "vqa_quality_wait", self.ctx)
        return out.reshape(n, npl)

    def quality(self, ref, dist, planes, ssim_mode=N.SSIM_GAUSS, frame_bytes=None):
        """SSE + SSIM per plane for n frame pairs; returns [n, n_planes] structured array (PLANE_DTYPE)."""
        self.quality_submit(ref, dist, planes, ssim_mode, frame_bytes)
        return self.quality_wait()

    # ---- per-kernel timing ---------------------------------------------------
    def profile(self, on=True):
        N.check(self.lib.vqa_profile_enable(self.ctx, 1 if on else 0), "vqa_profile_enable", self.ctx)

    def profile_read(self, reset=False):
        """-> {kernel name: (total_ms, launches)} for kernels launched since the last reset."""
        out = {}
        for k in range(N.K_COUNT):
            ms, cnt = C.c_double(0), C.c_int64(0)
            N.check(self.lib.vqa_profile_read(self.ctx, k, C.byref(ms), C.byref(cnt), 1 if reset else 0),
                    "vqa_profile_read", self.ctx)
            if cnt.value:
                out[self.lib.vqa_kernel_name(k).decode()] = (ms.value, cnt.value)
        return out

    # ---- debug -------------------------------------------------------------
    def debug_plane(self, which, frame, h, w):
        out = np.empty((h, w), np.uint8)
        st = self.lib.vqa_debug_read_plane(self.ctx, which, frame, out.ctypes.data, h, w)
        N.check(st, "vqa_debug_read_plane", self.ctx)
        return out
