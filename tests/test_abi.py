"""The C-ABI library loads and exports every symbol include/vqa.h declares (no compute calls)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(REPO, "include", "vqa.h")).read()
    return sorted(set(re.findall(r"^VQA_API[^;(]*?\b(vqa_\w+)\s*\(", txt, flags=re.M)))


def test_header_symbols_all_exported_and_bound():
    from rtvqa_amd import _native as N
    lib = N.load()
    names = _declared()
    assert len(names) >= 20
    assert set(names) == set(N.SIGNATURES), set(names) ^ set(N.SIGNATURES)
    for n in names:
        assert getattr(lib, n) is not None
    assert lib.vqa_abi_version() == N.VQA_ABI_VERSION


def test_struct_layouts_match_header():
    from rtvqa_amd import _native as N
    from rtvqa_amd.engine import FRAME_DTYPE, PLANE_DTYPE
    # ask the C compiler what include/vqa.h means
    import subprocess, tempfile
    src = r"""
#include <stdio.h>
#include <stddef.h>
#include "vqa.h"
int main(void){
  printf("%zu %zu %zu %zu\n", sizeof(vqa_params), sizeof(vqa_plane_desc), sizeof(vqa_plane_metrics), sizeof(vqa_frame_metrics));
  printf("%zu %zu %zu %zu %zu %zu %zu\n", offsetof(vqa_frame_metrics, hist_bgr), offsetof(vqa_frame_metrics, sum_gray2),
         offsetof(vqa_frame_metrics, dct_energy), offsetof(vqa_frame_metrics, sad_sum), offsetof(vqa_frame_metrics, mv_d2_hist),
         offsetof(vqa_frame_metrics, edge_count), offsetof(vqa_frame_metrics, has_prev));
  return 0; }
"""
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "l.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(REPO, "include"), "-o", os.path.join(d, "l"), os.path.join(d, "l.c")])
        out = subprocess.check_output([os.path.join(d, "l")]).decode().split()
    sizes, offs = list(map(int, out[:4])), list(map(int, out[4:]))
    assert sizes == [C.sizeof(N.VqaParams), C.sizeof(N.VqaPlaneDesc), C.sizeof(N.VqaPlaneMetrics), C.sizeof(N.VqaFrameMetrics)]
    assert offs == [getattr(N.VqaFrameMetrics, f).offset for f in
                    ("hist_bgr", "sum_gray2", "dct_energy", "sad_sum", "mv_d2_hist", "edge_count", "has_prev")]
    assert C.sizeof(N.VqaPlaneMetrics) == PLANE_DTYPE.itemsize
    assert C.sizeof(N.VqaFrameMetrics) == FRAME_DTYPE.itemsize
    for name in FRAME_DTYPE.names:
        assert FRAME_DTYPE.fields[name][1] == getattr(N.VqaFrameMetrics, name).offset, name


def test_constants_match_header():
    """Mask bits, modes, status codes and kernel ids of the ctypes binding are the header's."""
    from rtvqa_amd import _native as N
    txt = open(os.path.join(REPO, "include", "vqa.h")).read()
    for name, expr in re.findall(r"^#define\s+VQA_(M_\w+|DCT_\w+|SSIM_\w+|MOTION_\w+|ABI_VERSION)\s+(\(?[0-9xA-Fa-fuU <]+\)?)", txt, flags=re.M):
        val = eval(re.sub(r"(?<=[0-9a-fA-F])[uU]", "", expr))
        assert getattr(N, "VQA_ABI_VERSION" if name == "ABI_VERSION" else name) == val, name
    enum = re.search(r"enum vqa_kernel_id \{(.*?)\}", txt, flags=re.S).group(1)
    ids = dict((k, int(v)) for k, v in re.findall(r"VQA_(K_\w+)\s*=\s*(\d+)", enum))
    assert len(ids) >= 10
    for k, v in ids.items():
        assert getattr(N, k) == v, k
    lib = N.load()
    lib.vqa_kernel_name.restype = C.c_char_p
    names = [lib.vqa_kernel_name(i).decode() for i in range(N.K_COUNT)]
    assert len(set(names)) == N.K_COUNT and "?" not in names
    for k, v in re.findall(r"VQA_(ERR_\w+|OK)\s*=\s*(-?\d+)", txt):
        assert getattr(N, "VQA_" + k) == int(v), k


def test_error_paths_without_compute():
    from rtvqa_amd import _native as N
    lib = N.load()
    assert lib.vqa_strerror(N.VQA_ERR_NO_DEVICE).decode().startswith("no HIP device")
    assert lib.vqa_create(0, None) == N.VQA_ERR_INVALID
    p = N.VqaParams()
    lib.vqa_default_params(C.byref(p))
    assert (p.canny_low, p.canny_high, p.sad_range, p.dct_mode, p.resize_w) == (100, 200, 7, 0, 0)
    n = C.c_int(-1)
    lib.vqa_device_count(C.byref(n))
    if n.value == 0:
        # the product path must fail loudly when there is no GPU: no CPU fallback exists
        ctx = C.c_void_p()
        assert lib.vqa_create(0, C.byref(ctx)) == N.VQA_ERR_NO_DEVICE
        import rtvqa_amd
        with pytest.raises(N.VqaError):
            rtvqa_amd.Engine(0)
        with pytest.raises(N.VqaError):
            rtvqa_amd.complexity_metrics.process_dct_frame(np.zeros((16, 16, 3), np.uint8), 16, 16)


def test_shipped_library_has_no_hidden_switches():
    """The default library is the product: flavour 0, and the only environment name in its strings is the documented
    VQA_OVERLAP.  The lab build (superseded kernels, selectors, test seams) is a separate file that nothing loads by default."""
    import subprocess
    from rtvqa_amd import _native as N
    assert os.path.basename(N.LIB_PATH) == "libvqa_hip.so" or os.environ.get("VQA_LIB_PATH")
    shipped = os.path.join(REPO, "real-time-video-quality-analysis_amd", "csrc", "libvqa_hip.so")
    # (un-anchored: `strings` glues a printable byte that happens to precede a literal onto it)
    names = set(re.findall(r"VQA_[A-Z0-9_]+", subprocess.run(["strings", shipped], capture_output=True, text=True).stdout))
    assert names == {"VQA_OVERLAP"}, names
    hdr = open(os.path.join(REPO, "include", "vqa.h")).read()
    assert "VQA_OVERLAP" in hdr and "VQA_OPT_OVERLAP" in hdr
    lib = C.CDLL(shipped)
    assert lib.vqa_build_flavour() == 0
    lib.vqa_comm_debug_trace.restype = C.c_char_p
    assert lib.vqa_comm_debug_trace() == b""  # no stand-in in the shipped library
    lab = C.CDLL(N.LAB_LIB_PATH)
    assert lab.vqa_build_flavour() == (N.FLAVOUR_AB_VARIANTS | N.FLAVOUR_TEST_SEAMS)
    lab_names = set(re.findall(r"VQA_[A-Z0-9_]+", subprocess.run(["strings", N.LAB_LIB_PATH], capture_output=True, text=True).stdout))
    assert {"VQA_COMM_FAKE_RCCL", "VQA_HYST_MAX_ROUNDS", "VQA_HYST_RESCUE_MAX_ROUNDS", "VQA_FAIL_ENSURE_AT", "VQA_NMS_VARIANT", "VQA_DCT_VARIANT",
            "VQA_SAD_VARIANT", "VQA_SSIM_VARIANT"} <= lab_names


def test_product_never_imports_the_oracle():
    pkg = os.path.join(REPO, "real-time-video-quality-analysis_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                txt = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b|#include\s+\"[^\"]*oracle", txt, flags=re.M), f


def _build_demo(tmp_path, name="vqa_demo", lab=False):
    import subprocess
    exe = str(tmp_path / (name + ("_lab" if lab else "")))
    libdir = os.path.join(REPO, "real-time-video-quality-analysis_amd", "csrc", *(["lab"] if lab else []))
    subprocess.check_call(["gcc", "-O2", "-Wall", "-Werror", "-pthread", "-I", os.path.join(REPO, "include"), "-o", exe,
                           os.path.join(REPO, "examples", name + ".c"), "-L", libdir,
                           "-l:libvqa_hip_lab.so" if lab else "-lvqa_hip", "-Wl,-rpath," + libdir, "-lm"])
    return exe


def test_plain_c_host_links_against_the_abi(tmp_path):
    """examples/vqa_demo.c: the boundary is usable from C with nothing but include/vqa.h and the .so.
    Without a GPU the program must stop at vqa_create with the no-device error (no CPU fallback)."""
    import subprocess
    from rtvqa_amd import _native as N
    N.load()
    exe = _build_demo(tmp_path)
    n = C.c_int(-1)
    N.load().vqa_device_count(C.byref(n))
    if n.value == 0:
        r = subprocess.run([exe], capture_output=True, text=True)
        assert r.returncode == 2 and "no HIP device" in r.stderr


def test_multi_device_c_host_builds_and_refuses_to_run_without_a_device(tmp_path):
    """examples/vqa_multi.c (BASELINE configs[4] as a plain-C host: one thread + context per device, one all-reduce)."""
    import subprocess
    from rtvqa_amd import _native as N
    exe = _build_demo(tmp_path, "vqa_multi")
    n = C.c_int(-1)
    N.load().vqa_device_count(C.byref(n))
    if n.value == 0:
        r = subprocess.run([exe], capture_output=True, text=True)
        assert r.returncode == 2 and "no HIP device" in r.stderr


@pytest.mark.gpu
def test_multi_device_c_host_runs_with_the_devices_present(tmp_path):
    """On the one-GPU test box: 1 device, 6 x 2160p frames, 2 passes, then vqa_comm_create + vqa_allreduce over RCCL.
    (On a multi-GPU node the same binary takes every device; that run has not happened yet.)"""
    import subprocess
    r = subprocess.run([_build_demo(tmp_path, "vqa_multi"), "0", "6", "2"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "vqa_multi ok" in r.stdout, r.stdout + r.stderr
    assert re.search(r"all-reduce over \d+ device\(s\): \d+ frames", r.stdout) or "RCCL not installed" in r.stdout


@pytest.mark.gpu
def test_multi_device_c_host_rehearsal_three_workers_on_one_gpu(tmp_path):
    """The multi-device host for real, minus the second device: VQA_MULTI_REHEARSAL_DEVICE=0 puts three workers (three
    host threads, three contexts, three streams running concurrently) on the one GPU; linked against the LAB build, whose
    RCCL stand-in (VQA_COMM_FAKE_RCCL=1) lets vqa_comm_create take three contexts of one device.  The reduced frame count
    must be 3 x frames and every worker's self-checks (Parseval, hysteresis bound) must hold.  Against the SHIPPED library
    the same rehearsal must fail at vqa_comm_create: it has no stand-in and real RCCL takes one rank per device."""
    import subprocess
    env = dict(os.environ, VQA_MULTI_REHEARSAL_DEVICE="0", VQA_COMM_FAKE_RCCL="1")
    r = subprocess.run([_build_demo(tmp_path, "vqa_multi", lab=True), "3", "8", "2", "1080", "1920"], capture_output=True,
                       text=True, timeout=300, env=env)
    assert r.returncode == 0 and "vqa_multi ok" in r.stdout, r.stdout + r.stderr
    assert "all-reduce over 3 device(s): 24 frames" in r.stdout, r.stdout
    assert len(re.findall(r"^worker \d on device 0:", r.stdout, flags=re.M)) == 3
    r = subprocess.run([_build_demo(tmp_path, "vqa_multi"), "3", "2", "1", "270", "480"], capture_output=True, text=True,
                       timeout=300, env=env)
    assert r.returncode == 1 and "two contexts on device 0" in r.stderr, r.stdout + r.stderr


def test_streaming_c_host_builds_and_stops_without_a_device(tmp_path):
    """examples/vqa_stream.c - the one-pass pipeline (pinned ring, copier threads, two contexts, one upload for both halves)
    against include/vqa.h alone: compiles with -Wall -Werror; without a GPU it stops at vqa_create (no CPU fallback)."""
    import subprocess
    from rtvqa_amd import _native as N
    exe = _build_demo(tmp_path, "vqa_stream")
    n = C.c_int(-1)
    N.load().vqa_device_count(C.byref(n))
    if n.value == 0:
        r = subprocess.run([exe, "10", "64", "64", "2", "3", "2"], capture_output=True, text=True)
        assert r.returncode == 2 and "no HIP device" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("args", [("60", "270", "480", "5", "7", "3"), ("33", "96", "128", "1", "4", "2"),
                                  ("120", "1080", "1920", "10", "24", "6"), ("9", "64", "64", "10", "100", "1")])
def test_streaming_c_host_runs_on_the_gpu(tmp_path, args):
    """The plain-C streaming host on the GPU: pageable clip -> ring -> ONE upload per chunk -> quality of every frame and
    the complexity suite of every interval-th frame (halo frames across chunk seams, chunks that hold no sample, a clip too
    short to hold one); its self-checks (SSE of ref vs ref + 1, histogram mass, period-2 records) must hold."""
    import subprocess
    r = subprocess.run([_build_demo(tmp_path, "vqa_stream"), *args], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "vqa_stream ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_plain_c_host_runs_on_the_gpu(tmp_path):
    import subprocess
    from rtvqa_amd import _native as N
    N.load()
    r = subprocess.run([_build_demo(tmp_path)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "vqa_demo ok" in r.stdout, r.stdout + r.stderr


def test_collective_entry_points_reject_bad_arguments():
    """vqa_comm_* / vqa_allreduce argument checks need no device (and never load RCCL)."""
    from rtvqa_amd import _native as N
    lib = N.load()
    out = C.c_void_p()
    assert lib.vqa_comm_create(None, 1, C.byref(out)) == N.VQA_ERR_INVALID
    assert lib.vqa_comm_create_rank(None, None, 128, 2, 0, C.byref(out)) == N.VQA_ERR_INVALID
    assert lib.vqa_comm_unique_id(None, 128) == N.VQA_ERR_INVALID
    buf = (C.c_char * 64)()
    assert lib.vqa_comm_unique_id(buf, 64) == N.VQA_ERR_INVALID  # an id needs VQA_COMM_ID_BYTES = 128 bytes
    assert lib.vqa_allreduce(None, (C.c_double * 2)(), 2) == N.VQA_ERR_INVALID
    assert lib.vqa_comm_destroy(None) == N.VQA_ERR_INVALID


@pytest.mark.gpu
def test_allreduce_over_rccl_single_device():
    """The C-ABI collective on the one GPU of the test box: both ways of building a communicator (all local devices /
    join by rank) and a SUM all-reduce of pooled float64 scalars; more than 64 values or a duplicate device is refused."""
    from rtvqa_amd import _native as N
    lib = N.load()
    ctx = C.c_void_p()
    N.check(lib.vqa_create(0, C.byref(ctx)), "vqa_create")
    try:
        comm = C.c_void_p()
        arr = (C.c_void_p * 1)(ctx)
        st = lib.vqa_comm_create(arr, 1, C.byref(comm))
        if st == N.VQA_ERR_UNSUPPORTED:
            pytest.skip("librccl.so.1 is not installed")
        N.check(st, "vqa_comm_create", ctx)
        assert lib.vqa_comm_size(comm) == 1
        vals = (C.c_double * 7)(*[1.5, -2.25, 3e300, 4.0, 5.0, 6.0, 2257755.44])
        N.check(lib.vqa_allreduce(comm, vals, 7), "vqa_allreduce")
        assert list(vals) == [1.5, -2.25, 3e300, 4.0, 5.0, 6.0, 2257755.44]
        assert lib.vqa_allreduce(comm, (C.c_double * 65)(), 65) == N.VQA_ERR_INVALID
        N.check(lib.vqa_comm_destroy(comm), "vqa_comm_destroy")
        two = (C.c_void_p * 2)(ctx, ctx)
        assert lib.vqa_comm_create(two, 2, C.byref(comm)) == N.VQA_ERR_INVALID  # one ctx per device
        assert b"two contexts on device 0" in lib.vqa_comm_last_error(None)     # a failed creation says why
        uid = (C.c_char * 128)()
        N.check(lib.vqa_comm_unique_id(uid, 128), "vqa_comm_unique_id")
        N.check(lib.vqa_comm_create_rank(ctx, uid, 128, 1, 0, C.byref(comm)), "vqa_comm_create_rank")
        one = (C.c_double * 2)(7.0, 9.0)
        N.check(lib.vqa_allreduce(comm, one, 2), "vqa_allreduce")
        assert list(one) == [7.0, 9.0]
        N.check(lib.vqa_comm_destroy(comm), "vqa_comm_destroy")
    finally:
        lib.vqa_destroy(ctx)


_SEAM_SCRIPT = r"""
import ctypes as C, json, sys
sys.path.insert(0, %r)
from rtvqa_amd import _native as N
lib = N.load()
ctxs = []
for _ in range(3):
    c = C.c_void_p()
    N.check(lib.vqa_create(0, C.byref(c)), "vqa_create")
    ctxs.append(c)
comm = C.c_void_p()
arr = (C.c_void_p * 3)(*ctxs)
N.check(lib.vqa_comm_create(arr, 3, C.byref(comm)), "vqa_comm_create")
vals = (C.c_double * 6)(1.0, 2.0, 10.0, 20.0, 100.0, 200.0)   # [3 contexts][2]
N.check(lib.vqa_allreduce(comm, vals, 2), "vqa_allreduce")
size = lib.vqa_comm_size(comm)
N.check(lib.vqa_comm_destroy(comm), "vqa_comm_destroy")
for c in ctxs:
    lib.vqa_destroy(c)
print(json.dumps({"vals": list(vals), "size": size, "trace": lib.vqa_comm_debug_trace().decode()}))
"""


@pytest.mark.gpu
def test_multi_context_allreduce_through_the_test_seam():
    """vqa_comm_create with n_ctx > 1 (single process, one context per device: SURVEY 8e's ncclCommInitAll design) has
    never had two devices to run on.  In the LAB build, with VQA_COMM_FAKE_RCCL=1, an in-library stand-in replaces the RCCL entry points, so
    everything AROUND them runs on the one GPU of the test box: the device list handed to CommInitAll, one scratch buffer
    per context on its device, the H2D staging of each context's row, group start / one AllReduce per context / group
    end, the D2H of every row.  The stand-in refuses an AllReduce outside a group or a group that lacks a rank."""
    import json
    import subprocess
    import sys
    from rtvqa_amd import _native as N
    env = dict(os.environ, VQA_COMM_FAKE_RCCL="1", VQA_LIB_PATH=N.LAB_LIB_PATH)  # the stand-in exists in the lab build only
    out = subprocess.run([sys.executable, "-c", _SEAM_SCRIPT % REPO], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    got = json.loads(out.stdout.strip().splitlines()[-1])
    assert got["vals"] == [111.0, 222.0] * 3 and got["size"] == 3
    t = got["trace"]
    assert t.startswith("CommInitAll(n=3,devs=0,0,0);GroupStart;")
    assert [t.count("AllReduce(rank=%d,count=2,buf_dev=0);" % r) for r in range(3)] == [1, 1, 1]
    assert t.index("GroupStart;") < t.index("AllReduce(rank=0") < t.index("AllReduce(rank=2") < t.index("GroupEnd(3);")
    assert t.count("CommDestroy;") == 3
