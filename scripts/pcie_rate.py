"""PCIe-inclusive rate of workload c2: frames start in PINNED HOST memory every step (H2D inside the
submit call), ping-ponged over two contexts so copy and compute overlap.  Quoted in DESIGN.md §6; it is
never bench.py's `value`."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, rtvqa_amd
from rtvqa_amd import _native as N, synth
from rtvqa_amd.engine import bgr_planes

h, w, B, steps = 1080, 1920, 64, 8
engs = [rtvqa_amd.Engine(0), rtvqa_amd.Engine(0)]
fb = h * w * 3


def pinned(eng, n):
    p = C.c_void_p()
    N.check(eng.lib.vqa_alloc_pinned(eng.ctx, n, C.byref(p)), "pinned", eng.ctx)
    return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n,))


ref = synth.s_natural(B + 1, h, w, seed=3)
bufs = []
for e in engs:
    r = pinned(e, fb * (B + 1)).reshape(B + 1, h, w, 3); d = pinned(e, fb * (B + 1)).reshape(B + 1, h, w, 3)
    r[:] = ref; d[:] = synth.distort(ref)
    bufs.append((r, d))
planes = bgr_planes(h, w)
params = engs[0].make_params(dct_mode=N.DCT_BLOCK8)


def submit(i):
    e = engs[i & 1]; r, d = bufs[i & 1]
    e.quality_submit(r[1:], d[1:], planes, N.SSIM_GAUSS)
    e.complexity_submit(d[1:], d[0], N.M_DCT | N.M_TEMPORAL_DCT, params)


# "once": each buffer crosses PCIe ONCE per step (vqa_copy_h2d, async on the ctx stream) and both submits
# read the device copy; "submit" hands the pinned host pointers to both submits (dist crosses twice)
from rtvqa_amd.engine import DeviceBuffer, DeviceFrames
dbufs = [(DeviceBuffer(e, fb * (B + 1)), DeviceBuffer(e, fb * (B + 1))) for e in engs]


def submit_once(i):
    e = engs[i & 1]; r, d = bufs[i & 1]; dr, dd = dbufs[i & 1]
    N.check(e.lib.vqa_copy_h2d(e.ctx, dr.ptr, r.ctypes.data, r.nbytes), "h2d", e.ctx)
    N.check(e.lib.vqa_copy_h2d(e.ctx, dd.ptr, d.ctypes.data, d.nbytes), "h2d", e.ctx)
    fr, fd = DeviceFrames(dr.ptr, B + 1, h, w, owner=dr), DeviceFrames(dd.ptr, B + 1, h, w, owner=dd)
    e.quality_submit(fr.slice(1, B + 1), fd.slice(1, B + 1), planes, N.SSIM_GAUSS)
    e.complexity_submit(fd.slice(1, B + 1), fd.frame(0), N.M_DCT | N.M_TEMPORAL_DCT, params)


def wait(i):
    e = engs[i & 1]
    return e.quality_wait(), e.complexity_wait()


for mode in ("serial", "overlapped", "overlapped-once"):
    sub = submit_once if mode.endswith("once") else submit
    sub(0); wait(0)
    t0 = time.perf_counter()
    if mode == "serial":
        for i in range(steps):
            sub(0); wait(0)
    else:
        sub(0)
        for i in range(1, steps):
            sub(i); wait(i - 1)
        wait(steps - 1)
    dt = time.perf_counter() - t0
    gb = (2 if mode.endswith("once") else 3) * fb * (B + 1) * steps / 1e9
    print("%s: %.0f frames/s PCIe-inclusive, %.1f GB/s H2D" % (mode, B * steps / dt, gb / dt))
