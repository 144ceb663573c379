"""ctypes binding of include/vqa.h (libvqa_hip.so).

There is NO fallback: if the HIP library is missing or no gfx950 device is
visible, importing the library or creating a context raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# VQA_LIB_PATH: load another build of the same ABI (the lab build csrc/lab/libvqa_hip_lab.so for A/B re-measurement and
# fault injection, or a probe build); default = the shipped in-tree library
LIB_PATH = os.environ.get("VQA_LIB_PATH") or os.path.join(_HERE, "csrc", "libvqa_hip.so")
LAB_LIB_PATH = os.path.join(_HERE, "csrc", "lab", "libvqa_hip_lab.so")

VQA_ABI_VERSION = 7
VQA_TABLE_CACHE_GEOMETRIES = 16

VQA_OK = 0
VQA_ERR_INVALID = -1
VQA_ERR_NO_DEVICE = -2
VQA_ERR_HIP = -3
VQA_ERR_OOM = -4
VQA_ERR_UNSUPPORTED = -5
VQA_ERR_STATE = -6
VQA_ERR_INCOMPLETE = -7

VQA_MEM_HOST = 0
VQA_MEM_DEVICE = 1

M_GRAY_HIST = 1 << 0
M_COLOR_HIST = 1 << 1
M_DCT = 1 << 2
M_TEMPORAL_DCT = 1 << 3
M_EDGE = 1 << 4
M_MOTION = 1 << 5
M_ORB = 1 << 6
M_ALL = 0x7F

(K_GRAY_HIST, K_RESIZE, K_DCT8, K_DCT_FULL, K_CANNY_NMS, K_CANNY_HYST, K_SAD, K_SSIM_GAUSS, K_SSIM_FFMPEG, K_ORB,
 K_FARNEBACK, K_COUNT) = range(12)

OPT_OVERLAP, OPT_HYST_STATS = 0, 1
FLAVOUR_AB_VARIANTS, FLAVOUR_TEST_SEAMS = 1, 2

DCT_AUTO, DCT_BLOCK8, DCT_FULL = 0, 1, 2
SSIM_GAUSS, SSIM_FFMPEG = 0, 1
MOTION_SAD, MOTION_FARNEBACK = 0, 1


class VqaParams(C.Structure):
    _fields_ = [("resize_w", C.c_int32), ("resize_h", C.c_int32),
                ("canny_low", C.c_int32), ("canny_high", C.c_int32),
                ("sad_range", C.c_int32), ("dct_mode", C.c_int32),
                ("motion_mode", C.c_int32), ("reserved", C.c_int32 * 9)]


class VqaFrameMetrics(C.Structure):
    _fields_ = [("hist_gray", C.c_uint32 * 256),
                ("hist_bgr", (C.c_uint32 * 256) * 3),
                ("sum_gray2", C.c_uint64),
                ("dct_energy", C.c_double),
                ("temporal_dct_l1", C.c_double),
                ("sad_sum", C.c_uint64),
                ("sad_blocks", C.c_uint32),
                ("mv_d2_hist", C.c_uint32 * 129),
                ("edge_count", C.c_uint32),
                ("edge_strong", C.c_uint32),
                ("edge_weak", C.c_uint32),
                ("has_prev", C.c_uint32),
                ("hyst_steps", C.c_uint32), ("orb_keypoints", C.c_uint32), ("orb_response", C.c_uint32), ("hyst_overflow", C.c_uint32),
                ("flow_mag_mean", C.c_double)]


class VqaPlaneDesc(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32),
                ("offset", C.c_int64), ("row_stride", C.c_int64),
                ("pixel_step", C.c_int32), ("pad_", C.c_int32)]


class VqaPlaneMetrics(C.Structure):
    _fields_ = [("sse", C.c_uint64), ("ssim", C.c_double)]


# every symbol include/vqa.h declares: (restype, argtypes)
_u8p = C.c_void_p
SIGNATURES = {
    "vqa_abi_version": (C.c_int, []),
    "vqa_strerror": (C.c_char_p, [C.c_int]),
    "vqa_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "vqa_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "vqa_destroy": (C.c_int, [C.c_void_p]),
    "vqa_trim": (C.c_int, [C.c_void_p]),
    "vqa_last_hip_error": (C.c_char_p, [C.c_void_p]),
    "vqa_default_params": (None, [C.POINTER(VqaParams)]),
    "vqa_build_flavour": (C.c_int, []),
    "vqa_set_option": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "vqa_get_option": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "vqa_alloc_pinned": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "vqa_free_pinned": (C.c_int, [C.c_void_p, C.c_void_p]),
    "vqa_host_is_pinned": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_int)]),
    "vqa_alloc_device": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "vqa_free_device": (C.c_int, [C.c_void_p, C.c_void_p]),
    "vqa_copy_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "vqa_copy_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "vqa_sync": (C.c_int, [C.c_void_p]),
    "vqa_stream_wait": (C.c_int, [C.c_void_p, C.c_void_p]),
    "vqa_stream": (C.c_void_p, [C.c_void_p]),
    "vqa_complexity_submit": (C.c_int, [C.c_void_p, _u8p, _u8p, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_int64, C.c_int64, C.c_uint32, C.POINTER(VqaParams)]),
    "vqa_complexity_wait": (C.c_int, [C.c_void_p, C.POINTER(VqaFrameMetrics), C.c_int]),
    "vqa_quality_submit": (C.c_int, [C.c_void_p, _u8p, _u8p, C.c_int, C.c_int, C.c_int64, C.c_int64,
                                     C.POINTER(VqaPlaneDesc), C.c_int, C.c_int]),
    "vqa_quality_wait": (C.c_int, [C.c_void_p, C.POINTER(VqaPlaneMetrics), C.c_int]),
    "vqa_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "vqa_profile_read": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_int]),
    "vqa_kernel_name": (C.c_char_p, [C.c_int]),
    "vqa_debug_read_plane": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _u8p, C.c_int, C.c_int]),
    "vqa_comm_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_void_p)]),
    "vqa_comm_unique_id": (C.c_int, [C.c_void_p, C.c_size_t]),
    "vqa_comm_create_rank": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "vqa_comm_destroy": (C.c_int, [C.c_void_p]),
    "vqa_comm_size": (C.c_int, [C.c_void_p]),
    "vqa_comm_last_error": (C.c_char_p, [C.c_void_p]),
    "vqa_comm_debug_trace": (C.c_char_p, []),
    "vqa_allreduce": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.c_int]),
}

_lib = None


class VqaError(RuntimeError):
    def __init__(self, status, where, detail=""):
        self.status = status
        msg = "%s failed: %s (%d)" % (where, _strerror(status), status)
        if detail:
            msg += " — " + detail
        super().__init__(msg)


def _strerror(status):
    try:
        return load().vqa_strerror(status).decode()
    except Exception:
        return "status %d" % status


def load():
    """Load libvqa_hip.so and bind every declared symbol.  Raises if the
    library has not been built (run `python -c 'import __graft_entry__ as g; g.build()'`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "HIP extension not built: %s is missing. Build it with `make -C %s` "
            "(there is no CPU fallback)." % (LIB_PATH, os.path.dirname(LIB_PATH)))
    # PyTorch-ROCm bundles its own libamdhip64.so.7.  If torch is importable, load it
    # first so this process holds ONE HIP runtime (the dynamic linker then resolves
    # our DT_NEEDED libamdhip64.so.7 to the copy torch already mapped).
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL if hasattr(C, "RTLD_GLOBAL") else 0)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.vqa_abi_version() != VQA_ABI_VERSION:
        raise ImportError("libvqa_hip.so ABI %d != binding ABI %d" % (lib.vqa_abi_version(), VQA_ABI_VERSION))
    _lib = lib
    return lib


def check(status, where, ctx=None):
    if status != VQA_OK:
        detail = ""
        if ctx is not None and status in (VQA_ERR_HIP, VQA_ERR_OOM, VQA_ERR_INCOMPLETE):
            detail = load().vqa_last_hip_error(ctx).decode()
        raise VqaError(status, where, detail)


# ---------------------------------------------------------------------------
# roctx ranges (SURVEY.md section 5 "Tracing": rocprofv3 --marker-trace): named ranges around the host-side stages of a pass,
# so a trace shows gather / upload / submit / wait / tails per chunk and lane next to the kernels and copies.  The marker
# library is looked for only when a profiler is attached (rocprofv3 preloads its tool library) or VQA_ROCTX=1 asks for it;
# everywhere else - and when the library is absent - trace_range() is a no-op that costs one attribute test.
# ---------------------------------------------------------------------------
_roctx = None


def _roctx_wanted():
    v = os.environ.get("VQA_ROCTX")
    if v is not None:
        return v not in ("", "0")
    return "rocprofiler" in os.environ.get("LD_PRELOAD", "") or "ROCP_TOOL_LIBRARIES" in os.environ


def _roctx_load():
    global _roctx
    if _roctx is None:
        _roctx = False
        if _roctx_wanted():
            for name in ("librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"):
                for where in ("", "/opt/rocm/lib/"):
                    try:
                        lib = C.CDLL(where + name)
                        lib.roctxRangePushA.argtypes = [C.c_char_p]
                        lib.roctxRangePushA.restype = C.c_int
                        lib.roctxRangePop.restype = C.c_int
                        _roctx = lib
                        return _roctx
                    except (OSError, AttributeError):
                        continue
    return _roctx


class _Range:
    __slots__ = ("name",)

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        _roctx.roctxRangePushA(self.name)

    def __exit__(self, *a):
        _roctx.roctxRangePop()
        return False


class _NoRange:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


_NO_RANGE = _NoRange()


def trace_range(name, *args):
    """with trace_range("vqa:upload k=%d lane=%d", k, lane): ... - a roctx range when markers are on, else nothing"""
    lib = _roctx if _roctx is not None else _roctx_load()
    if not lib:
        return _NO_RANGE
    return _Range((name % args if args else name).encode())


def roctx_active():
    return bool(_roctx if _roctx is not None else _roctx_load())
