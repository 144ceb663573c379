"""One pass of a clip through the engine: every byte crosses PCIe once, every frame is measured once.

This is the host side of the reference's hot path as its callers drive it:

    process_video_and_extract_metrics   video_processing.py:216 (run_ffmpeg_metrics) + :242
                                        (calculate_average_scene_complexity) on the SAME encoded stream
    run_ffmpeg_metrics                  video_processing.py:270-297   (quality only)
    calculate_average_scene_complexity  complexity_metrics.py:246-310 (complexity only)

The reference decodes the encoded stream three times and pickles every frame into a process pool seven times
(SURVEY.md 3.2); here a clip is cut into chunks, a chunk is brought to the device ONCE and both the quality kernels
(all frames) and the complexity kernels (every frame_interval-th frame, a strided view of the same device bytes)
read it there.  Where the frames come from decides how they travel:

    device    DeviceFrames (torch tensor, vqa_alloc_device): views, nothing moves
    pinned    page-locked host memory (Engine.alloc_pinned, torch pin_memory - north_star's "decoded frames in
              pinned buffers"): vqa_copy_h2d straight from the caller's memory, an asynchronous DMA
    pageable  ordinary NumPy arrays / np.load(mmap_mode="r"): copier threads move chunk k+1 into a slot of a
              pinned ring (np.copyto releases the GIL) while chunk k crosses PCIe and chunk k-1 is on the GPU

Every upload of a pass goes through the device's COPY LANE (get_copy_engine: one more engine, whose stream carries nothing
but H2D copies), so chunks cross PCIe one after another, in order; chunks alternate between two measuring engines (= two
HIP streams with their own scratch), each of which waits ON THE DEVICE for its chunk's upload (vqa_stream_wait).  The
upload of chunk k+1 therefore runs under the kernels of chunk k, the copiers gather chunk k+2 meanwhile, and the host-side
float tails (tails.py) of a finished chunk run while the GPU works on the next.  Nothing here computes a metric: pointers
in, records out.
"""
import os
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import _native as N
from . import tails
from .engine import DeviceBuffer, DeviceFrames, Engine
from .pooling import shard_range

# A chunk is at most batch_size frames and, when it travels over PCIe, at most this many bytes (all streams together).  With the
# copy lane (round 6) the link is busy from the first chunk's upload to the last one's, so what a chunk size still decides is the
# exposed tail - the last chunk's kernels run after the link has gone quiet - against the per-chunk host time.  Measured on 257 x
# 1080p pairs, frames/s at 64 / 128 / 160 / 256 / 512 / 1024 MiB - pinned: 3911 / 4144 / 4209 / 4296 / 4290 / 4246; pageable (the
# first gather overlaps nothing, so short chunks also start the pipeline sooner): 3873 / 3790 / 4090 / 4147 / 3997 / 3753
# (round 5, uploads on the lanes' own streams: pinned best at 1 GiB with 4064, pageable at 160-384 MiB with 4173)
# Big frames: a chunk is also at least CHUNK_FRAMES_MIN frames (while that stays below CHUNK_BYTES_HARD) - per-chunk host time is
# per chunk, not per byte: 65 x 2160p pairs (49.8 MB each), same caps - pinned: 728 / 880 / 945 / 999 / 1052 / 1070, pageable:
# 718 / 862 / 918 / 961 / 972 / 939 (256 MiB is 5 such pairs; 16 of them are 796 MB)
CHUNK_BYTES_MAX = 256 << 20
STAGED_CHUNK_BYTES_MAX = 256 << 20
CHUNK_FRAMES_MIN = 16
SAMPLES_PER_CHUNK = True      # (False: a chunk of a fused pass is batch_size source frames, as in round 5)
CHUNK_BYTES_HARD = 1 << 30
STAGE_THREADS = max(1, min(8, (os.cpu_count() or 2) // 2))  # copier threads of the pinned ring
MAX_LANES = 2                 # engines a pass alternates its chunks between (1: everything on the default engine, in order)
# measuring engines of a pass with Farneback motion.  Rounds 3-5 kept ONE (GiB-sized scratch per context, and two 64-frame
# batches in flight thrashed in round 3); with chunks of <= 256 MiB the scratch is 1-3 GB per context and the second engine hides
# the host time of a Farneback submit (~60 launches) behind the other chunk's kernels - c3ref's clip through the entry point, 1 / 2
# engines: resident 4.65-4.80 k / 4.71-4.75 k, pinned host 2.45-2.48 k / 2.93 k, pageable 2.17-2.20 k / 2.65-2.67 k frames/s
FARNEBACK_LANES = 2

KINDS = ("motion", "dct", "hist", "edge", "orb", "color")
MASK = {"dct": N.M_DCT, "temporal": N.M_TEMPORAL_DCT, "hist": N.M_GRAY_HIST, "color": N.M_COLOR_HIST,
        "edge": N.M_EDGE, "motion": N.M_MOTION, "orb": N.M_ORB}

_engines = {}
_second = {}
_copy = {}
_staging = {}
_lock = threading.RLock()


def get_engine(device=None):
    """One Engine per (process, device).  Default device: VQA_DEVICE, else LOCAL_RANK, else 0."""
    if device is None:
        device = int(os.environ.get("VQA_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    with _lock:
        if device not in _engines:
            _engines[device] = Engine(device)
        return _engines[device]


def get_engine_lanes(device=None, count=2):
    """`count` engines (= HIP streams with their own scratch) on one device: while chunk k's kernels run on one,
    chunk k+1 crosses PCIe on the next.  The first is the device's default engine."""
    first = get_engine(device)
    with _lock:
        more = _second.setdefault(first.device, [])
        while len(more) < count - 1:
            more.append(Engine(first.device))
        return (first,) + tuple(more[:count - 1])


def get_engine_pair(device=None):
    return get_engine_lanes(device, 2)


def get_copy_engine(device=None):
    """The device's COPY LANE: an engine that never measures anything - every upload of a pass is enqueued on its stream, so
    chunks cross PCIe one after another, in order, and the engine that measures a chunk waits for it on the device
    (Engine.wait_for = vqa_stream_wait).  Uploads enqueued on the measuring engines' own streams all start at once, share the
    link and finish together, and every chunk's kernels wait for all of them: rocprofv3 showed 1.4 % of the H2D time under a
    kernel that way (profiles/round6_api_trace_pinned.json, "before")."""
    first = get_engine(device)
    with _lock:
        if first.device not in _copy:
            _copy[first.device] = Engine(first.device)
        return _copy[first.device]


class _Staging:
    """What a device's passes keep between calls so that a steady stream of clips allocates nothing: the pinned ring
    and the per-lane device buffers (release_buffers() gives them back).  Owned by the process's default engine of the
    device, so a caller's short-lived engine can run a pass without taking the buffers with it.  Like the engines
    themselves this is per process and device and NOT thread-safe: one pass at a time per device."""

    def __init__(self, engine):
        self.engine = engine          # owner of the pinned slots and lane buffers
        self.slots = []               # pinned uint8 arrays, all of slot_bytes
        self.slot_bytes = 0
        self.dev = {}                 # (lane, name) -> DeviceBuffer
        self.pool = None

    def ring(self, count, nbytes):
        if nbytes > self.slot_bytes or len(self.slots) < count:
            for a in self.slots:
                self.engine.free_pinned(a)
            self.slots = []
            want = max(nbytes, self.slot_bytes)
            self.slot_bytes = 0
            fresh = []
            try:
                for _ in range(count):
                    fresh.append(self.engine.alloc_pinned((want,)))
            except BaseException:  # a half-built ring is given back whole: nothing leaks, the next pass starts from an empty ring
                for a in fresh:
                    self.engine.free_pinned(a)
                raise
            self.slots, self.slot_bytes = fresh, want
        return self.slots[:count]

    def device_buffer(self, lane, name, nbytes):
        b = self.dev.get((lane, name))
        if b is None or b.nbytes < nbytes:
            if b is not None:
                b.free()  # (no pass is running: every chunk of the previous one was waited for)
            b = self.dev[(lane, name)] = DeviceBuffer(self.engine, nbytes)
        return b

    def copiers(self):
        if self.pool is None:
            self.pool = ThreadPoolExecutor(max_workers=STAGE_THREADS, thread_name_prefix="vqa-stage")
        return self.pool

    def release(self):
        for a in self.slots:
            self.engine.free_pinned(a)
        self.slots, self.slot_bytes = [], 0
        for b in self.dev.values():
            b.free()
        self.dev = {}
        if self.pool is not None:
            self.pool.shutdown(wait=True)
            self.pool = None


def _staging_of(engine):
    with _lock:
        st = _staging.get(engine.device)
        if st is None:
            st = _staging[engine.device] = _Staging(get_engine(engine.device))
        return st


def release_buffers(device=None):
    """Give back what the passes keep between calls on `device` (all devices when None): the pinned ring, the lane
    buffers on the device and - vqa_trim - every engine's scratch and cached tables.  The reference holds nothing
    between calls (a process pool per call, complexity_metrics.py:143-147); the next call re-grows what it needs.
    Waits for a pass that is running on the device (pass_lock)."""
    with _lock:
        devs = [device] if device is not None else sorted(set(_staging) | set(_engines))
    for d in devs:
        with pass_lock(d):
            with _lock:
                st = _staging.pop(d, None)
                engs = ([_engines[d]] if d in _engines else []) + list(_second.get(d, ())) + ([_copy[d]] if d in _copy else [])
            if st is not None:
                st.release()
            for e in engs:
                e.trim()


# ---------------------------------------------------------------------------
# sources
# ---------------------------------------------------------------------------
class _Source:
    """n frames of `fb` bytes each: a device-resident clip or a host array whose first axis is the frame."""

    def __init__(self, frames, engine):
        self.frames = frames
        if isinstance(frames, DeviceFrames):
            self.kind, self.n = "device", frames.n
            self.fb = frames.h * frames.w * frames.channels
            return
        if frames.dtype != np.uint8:  # a silent cast (or, from pinned memory, a reinterpretation) would measure garbage
            raise ValueError("frames must be uint8 (got %s): decoded 8-bit frames, as cv2.VideoCapture.read yields" % frames.dtype)
        self.n = frames.shape[0]
        self.fb = int(np.prod(frames.shape[1:], dtype=np.int64)) if frames.ndim > 1 else 1
        compact = self.n == 0 or (frames[0].flags.c_contiguous and (self.n == 1 or frames.strides[0] >= self.fb))
        # padded rows / a region of interest inside larger frames: the ring compacts them on the way
        self.kind = "pinned" if (compact and self.n and engine.is_pinned(frames)) else "pageable"

    def view(self, start, count, step=1):
        return self.frames[start:start + (count - 1) * step + 1:step]


class _Copy:
    """frames `view` of a source -> frame slots [slot, slot + len(view)) of a chunk buffer"""

    def __init__(self, slot, view):
        self.slot, self.view = slot, view


def _fill_slot(pool, ring_frames, copies):
    """pageable -> pinned: every copy split over the copier threads by frames; returns the futures"""
    futs = []
    for cp in copies:
        n = len(cp.view)
        parts = min(STAGE_THREADS, n)
        for t in range(parts):
            a, b = n * t // parts, n * (t + 1) // parts
            dst = ring_frames[cp.slot + a:cp.slot + b].reshape((b - a,) + cp.view.shape[1:])  # (the ring is contiguous: a view)
            futs.append(pool.submit(np.copyto, dst, cp.view[a:b]))
    return futs


def _upload(engine, dev_ptr, fb, copies, ring_frames=None):
    """enqueue the H2D copies of a chunk: from the ring slot (contiguous spans) or straight from pinned user memory"""
    for cp in copies:
        n = len(cp.view)
        if ring_frames is not None:
            engine.h2d_async(dev_ptr + cp.slot * fb, ring_frames[cp.slot:].ctypes.data, n * fb)
        elif n == 1 or cp.view.strides[0] == fb:
            engine.h2d_async(dev_ptr + cp.slot * fb, cp.view.ctypes.data, n * fb)
        else:  # every k-th frame of a pinned clip: only the selected frames cross PCIe
            base, step = cp.view.ctypes.data, cp.view.strides[0]
            for i in range(n):
                engine.h2d_async(dev_ptr + (cp.slot + i) * fb, base + i * step, fb)


def stage_frames_start(engine, items, slot=0, slots=1, min_bytes=0):
    """Start gathering a list of equally shaped uint8 frames into ONE pinned array [n, ...] (process_in_batches' chunk:
    "whole batch staged in one pinned buffer and dispatched as one launch") - straight into slot `slot` of
    `slots` of the device's pinned ring, by the copier threads, no intermediate np.stack.  -> (array, futures); the array is
    complete once every future has a result and stays valid until the slot is staged into again.  The ring is re-allocated
    when a slot is too small - never while another slot is in use: a caller that alternates slots passes the largest size it
    will need as min_bytes on every call."""
    shape = items[0].shape
    for f in items:
        if f.dtype != np.uint8:  # a silent cast would turn float frames in 0..1 into all-zero planes
            raise ValueError("frames must be uint8 (got %s): decoded 8-bit BGR frames, as cv2.VideoCapture.read yields" % f.dtype)
        if f.shape != shape:
            raise ValueError("the frames of one launch must share a geometry (%s vs %s)" % (f.shape, shape))
    fb = int(np.prod(shape, dtype=np.int64))
    st = _staging_of(engine)
    blk = st.ring(slots, max(len(items) * fb, int(min_bytes)))[slot]
    out = blk[:len(items) * fb].reshape((len(items),) + shape)
    pool = st.copiers()
    parts = min(STAGE_THREADS, len(items))

    def gather(a, b):
        for i in range(a, b):
            np.copyto(out[i], items[i])
    return out, [pool.submit(gather, len(items) * t // parts, len(items) * (t + 1) // parts) for t in range(parts)]


def stage_frames(engine, items):
    """stage_frames_start, waited for: valid until the next pass / staging call on the device."""
    out, futs = stage_frames_start(engine, items)
    for f in futs:
        f.result()
    return out


# ---------------------------------------------------------------------------
# the pass
# ---------------------------------------------------------------------------
def selected_indices(num_frames, frame_interval):
    """0-based indices read_frame_pairs keeps: 1-based count % interval == 0 (complexity_metrics.py:103-104)."""
    return np.arange(frame_interval - 1, num_frames, frame_interval)


def plan_chunks(n, want_q, interval, lo, hi, cap, split=False):
    """The chunks of a pass, as pure arithmetic (no engine, no frames): -> list of dicts.

    n frames; quality wanted or not; complexity samples lo..hi-1 of the series at `interval` (None: no complexity), where
    sample j measures selected frame idx[1 + j] against idx[j], idx = selected_indices(n, interval); `cap` frames per chunk.
    A chunk's frames sit in a buffer of cap + 1 frame slots, slot 0 being the halo (the frame before the chunk's first
    sample when it lies before the chunk).  Per chunk:
      q0, qn        quality: source frames q0 .. q0 + qn - 1, in slots qslot .. qslot + qn - 1     (quality passes)
      j0, j1        complexity samples j0 .. j1 - 1; their frames sit in slots 1 + first + i * step, prev0 in slot
                    1 + prev_slot (prev_slot = -1: the halo slot)
      copies        (slot, first source frame, count, source step) of the distorted stream; rcopies: of the reference
    With quality the chunk is the dense source range [q0, q0 + qn) and the samples are every interval-th frame of it;
    without, the chunk holds exactly the samples' frames (compact), so only selected frames ever move.
    split: the quality kernels read their OWN pair of streams (planar yuv420p, what FFmpeg's filters compare,
    video_processing.py:274-276) and the complexity kernels the encoded stream as cv2 decodes it (BGR,
    complexity_metrics.py:100): `rcopies` / `qcopies` are the dense range of the quality pair (slots 0 .. qn - 1, qslot = 0)
    and `copies` the chunk's samples of the BGR stream, compact - every byte of either stream moves once."""
    plans = []
    if want_q:
        idx = selected_indices(n, interval) if interval else None
        for k, a in enumerate(range(0, n, cap)):
            b = min(a + cap, n)
            p = dict(k=k, q0=a, qn=b - a, qslot=0 if split else 1, j0=0, j1=0, copies=[], rcopies=[(0, a, b - a, 1)])
            if split:
                p["qcopies"] = [(0, a, b - a, 1)]
            if interval and hi > lo:
                # samples whose frame lies in [a, b): sample j measures selected frame idx[1 + j]
                j0 = max(lo, int(np.searchsorted(idx, a)) - 1, 0)
                j1 = max(j0, min(hi, int(np.searchsorted(idx, b)) - 1))
                p.update(j0=j0, j1=j1)
                if j1 > j0 and split:
                    p.update(first=0, step=1, prev_slot=-1)
                    if interval == 1:
                        p["copies"].append((0, int(idx[j0]), j1 - j0 + 1, 1))
                    else:
                        p["copies"] += [(0, int(idx[j0]), 1, 1), (1, int(idx[1 + j0]), j1 - j0, interval)]
                elif j1 > j0:
                    prev = int(idx[j0])
                    p.update(first=int(idx[1 + j0]) - a, step=interval, prev_slot=(prev - a) if prev >= a else -1)
                    if prev < a:
                        p["copies"].append((0, prev, 1, 1))
            if not split:
                p["copies"].append((1, a, b - a, 1))
            plans.append(p)
        return plans
    idx = selected_indices(n, interval)
    for k, j0 in enumerate(range(lo, hi, cap)):
        j1 = min(j0 + cap, hi)
        p = dict(k=k, j0=j0, j1=j1, first=0, step=1, prev_slot=-1, rcopies=[])
        if interval == 1:  # the frame before the first sample and the samples are one contiguous range
            p["copies"] = [(0, int(idx[j0]), j1 - j0 + 1, 1)]
        else:
            p["copies"] = [(0, int(idx[j0]), 1, 1), (1, int(idx[1 + j0]), j1 - j0, interval)]
        plans.append(p)
    return plans


def chunk_frames(batch_size, fused_interval, host_bytes_per_frame, staged):
    """Frames (complexity-only passes: samples) per chunk - pure arithmetic.

    batch_size is the reference's: items per process_in_batches chunk = SELECTED frames (complexity_metrics.py:128, :268-290).  A
    chunk of a FUSED pass (quality + complexity: fused_interval = its frame_interval, else None) is a dense source range, so it
    spans batch_size * interval frames: with config.json's interval 10 a launch measures up to 100 samples, not 10 (Farneback's
    pyramid runs at 0.42 of HBM on 64 pairs and 0.28 on 10).  A chunk that crosses PCIe (host_bytes_per_frame > 0: what one
    source frame brings along, all streams together) is also at most CHUNK_BYTES_MAX (STAGED_...: through the ring) - but at
    least CHUNK_FRAMES_MIN frames while that stays below CHUNK_BYTES_HARD."""
    by_batch = int(batch_size) * (fused_interval if (fused_interval and fused_interval > 1 and SAMPLES_PER_CHUNK) else 1)
    if host_bytes_per_frame <= 0:
        return max(1, by_batch)
    limit = STAGED_CHUNK_BYTES_MAX if staged else CHUNK_BYTES_MAX
    by_bytes = max(limit // host_bytes_per_frame, min(CHUNK_FRAMES_MIN, CHUNK_BYTES_HARD // host_bytes_per_frame))
    return max(1, min(by_batch, by_bytes))


class Complexity:
    """What the complexity half of a pass measures (arguments of calculate_average_scene_complexity)."""

    def __init__(self, resize, frame_interval=10, mask=N.M_ALL, dct_mode=N.DCT_AUTO, motion_mode=N.MOTION_SAD, shard=None):
        self.resize, self.interval, self.mask = resize, int(frame_interval), mask
        self.dct_mode, self.motion_mode, self.shard = dct_mode, motion_mode, shard


class Quality:
    """What the quality half compares: `planes` of every frame (engine.bgr_planes / yuv420p_planes ...)."""

    def __init__(self, planes, ssim_mode=N.SSIM_GAUSS):
        self.planes, self.ssim_mode = planes, ssim_mode


class _Feed:
    """One stream of a pass on its way to the kernels: its source and, for a host source, the lane buffers and the region
    of a ring slot its chunks travel through.  `key` names the plan's copy list for it."""

    def __init__(self, name, key, frames, engine):
        self.name, self.key = name, key
        self.src = _Source(frames, engine)
        self.fb = self.src.fb
        self.host = self.src.kind != "device"
        self.staged = self.src.kind == "pageable"
        self.slots = 0      # frame slots a chunk of this feed needs (from the plans)
        self.ring_off = 0   # byte offset of the feed's region inside a ring slot

    def geometry(self):
        f = self.src.frames
        return (f.h, f.w, f.channels) if isinstance(f, DeviceFrames) else tuple(f.shape[1:])


# One pass at a time per device: the pinned ring, the lane buffers and the engines' pending state are per device, and the
# reference's surface is written for threaded callers (video_processing.py:25-41 queue logging, :41,:62-67 CSV lock).  Every
# entry that touches them (run, process_in_batches' kernel branch, the per-frame callables, release_buffers) holds this.
_pass_locks = {}


def pass_lock(device=None):
    """The re-entrant lock that serialises passes on `device` (default device when None)."""
    if device is None:
        device = int(os.environ.get("VQA_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    with _lock:
        lk = _pass_locks.get(device)
        if lk is None:
            lk = _pass_locks[device] = threading.RLock()
        return lk


def run(dist, ref=None, quality=None, complexity=None, batch_size=100, engine=None, on_quality=None, qdist=None, device=None):
    """One pass over a clip.

    dist        the stream the complexity half reads: [N,H,W,3] uint8 BGR (host array, memmap or DeviceFrames); without
                `qdist` the quality half reads it too (for a quality-only pass any [N, frame_bytes] layout the planes describe)
    ref         the reference stream of the quality half (same layout as the distorted stream it is compared with), or None
    qdist       the distorted stream AS THE QUALITY FILTERS SEE IT when that is not `dist`: planar yuv420p [N, H*W*3/2] next to
                the encoded BGR stream (the reference compares decoded planes with FFmpeg, video_processing.py:274-276, and
                measures complexity on cv2's BGR decode of the same encoded file, :242-247).  Still ONE pass: per chunk the
                quality pair and the chunk's selected BGR frames are uploaded, each byte once
    quality     Quality or None;  complexity  Complexity or None
    on_quality  optional callback(first_frame, sse [m,p], ssim [m,p]) per finished chunk, in frame order (stats
                lines are formatted while the GPU works on the next chunk)
    device      the device of the default engine when `engine` is None and no stream is resident (config key "device")
    -> (sse [n,p] uint64, ssim [n,p] float64) or None, series dict or None.
    series: kind -> list in the reference's sample order (motion/dct/hist/edge/orb/color: T-1 samples, temporal:
    T-2, complexity_metrics.py:268-290, :533-537) and "range" = the shard's place in the whole series.
    Streams may live in different places (device / pinned / pageable): each travels its own way.  Passes on one device
    are serialised (pass_lock): a second thread's call waits for the first to finish."""
    want_q, want_c = quality is not None, complexity is not None
    if qdist is not None and not (want_q and want_c):
        if want_q:          # a quality-only pass compares the quality pair; the BGR stream has no reader
            dist = qdist
        qdist = None
    split = qdist is not None
    qd = qdist if split else dist       # the distorted stream of the quality half

    def count(fr):
        return fr.n if isinstance(fr, DeviceFrames) else fr.shape[0]

    on_device = isinstance(dist, DeviceFrames)
    if want_c and (dist.channels != 3 if on_device else (dist.ndim != 4 or dist.shape[3] != 3)):
        raise ValueError("frames must be uint8 [N,H,W,3] packed BGR")
    n = count(qd) if want_q else count(dist)
    if split and count(dist) != n:
        raise ValueError("the quality pair and the encoded stream must have the same number of frames (%d vs %d)" % (n, count(dist)))
    series = None
    if want_c:
        cx = complexity
        idx = selected_indices(n, cx.interval)
        T1 = max(len(idx) - 1, 0)
        lo, hi = shard_range(T1, *cx.shard) if cx.shard is not None else (0, T1)
        if cx.shard is not None and want_q:
            raise ValueError("a sharded pass measures complexity only")
        series = {k: [] for k in KINDS + ("temporal",)}
        series["range"] = (lo, hi)
        if hi <= lo:
            want_c = False  # unopenable / too short: empty series, as the reference's empty pair list (:95-97)
    if not want_q and not want_c:
        return None, series  # (no engine is created for a clip with nothing to measure)
    if n == 0:
        e = np.zeros((0, len(quality.planes)))
        return (e.astype(np.uint64), e), series
    if engine is not None:
        first = engine
    else:
        dev = next((d for d in (_device_of(f) for f in (dist, ref, qdist) if f is not None) if d is not None), device)
        first = get_engine(dev)
    with pass_lock(first.device):
        return _run_locked(first, engine, dist, ref, qd, split, quality, complexity if want_c else None, series, n,
                           batch_size, on_quality)


def _run_locked(first, engine, dist, ref, qd, split, quality, complexity, series, n, batch_size, on_quality):
    want_q, want_c = quality is not None, complexity is not None
    feeds = {}
    if want_c or not split:
        feeds["dist"] = _Feed("dist", "copies", dist, first)
    if want_q:
        feeds["ref"] = _Feed("ref", "rcopies", ref, first)
        if split:
            feeds["qdist"] = _Feed("qdist", "qcopies", qd, first)
        fq, fr = feeds["qdist" if split else "dist"], feeds["ref"]
        if fr.src.n != fq.src.n or fr.fb != fq.fb:
            raise ValueError("reference and distorted streams must have the same frame count and layout")
        if fr.geometry() != fq.geometry() and not (fr.host != fq.host):
            # (an equal-byte reshape - 1080x1920 against 1920x1080 - would be compared plane against garbage)
            raise ValueError("reference and distorted streams must share a geometry (%s vs %s)" % (fr.geometry(), fq.geometry()))
    if want_c:
        lo, hi = series["range"]
        idx = selected_indices(n, complexity.interval)
    else:
        lo = hi = 0

    # ---- chunks: (a, b) dense source range (quality present) or (j0, j1) sample range (complexity only)
    interval = complexity.interval if want_c else None
    per_frame = 0  # host bytes a source frame brings along
    if want_q:
        per_frame += sum(feeds[k].fb for k in (("ref", "qdist") if split else ("ref", "dist")) if feeds[k].host)
        if split and want_c and feeds["dist"].host:
            per_frame += -(-feeds["dist"].fb // interval)
    elif feeds["dist"].host:
        per_frame = feeds["dist"].fb
    host = any(f.host for f in feeds.values())
    staged = any(f.staged for f in feeds.values())
    cap = chunk_frames(batch_size, interval if (want_q and want_c) else None, per_frame if host else 0, staged)
    plans = plan_chunks(n, want_q, interval, lo, hi, cap, split)
    nchunks = len(plans)
    # ---- lanes
    farneback = want_c and (complexity.mask & N.M_MOTION) and complexity.motion_mode == N.MOTION_FARNEBACK
    want_lanes = min(MAX_LANES, FARNEBACK_LANES) if farneback else MAX_LANES
    if engine is not None or nchunks <= 1 or want_lanes < 2:
        lanes = [first]
    else:
        lanes = list(get_engine_lanes(first.device, min(want_lanes, nchunks)))
    st = _staging_of(first) if host else None
    cp = get_copy_engine(first.device) if host else None   # the copy lane: every upload of the pass, in chunk order
    nb = len(lanes) + 1    # buffer sets on the device: the chunks the lanes hold + the one whose upload runs under their kernels
    params = first.make_params(resize=complexity.resize, dct_mode=complexity.dct_mode,
                               motion_mode=complexity.motion_mode) if want_c else None
    # ---- buffers: every host feed gets a buffer per set (slot 0 of the complexity stream is the halo, the frame before the
    # chunk's first sample), every staged feed a region of each ring slot
    off = 0
    for f in feeds.values():
        f.slots = max([slot + cnt for p in plans for slot, _s, cnt, _t in p.get(f.key, ())] or [0])
        if f.host and f.slots:
            for bs in range(min(nb, nchunks)):
                st.device_buffer(bs, f.name, f.slots * f.fb)
        if f.staged:
            f.ring_off = off
            off += f.slots * f.fb
    # ring slots: the chunks the lanes hold (a slot is free again when its chunk has been waited for), the chunk being
    # uploaded and the chunk the copiers gather meanwhile
    ring = st.ring(min(len(lanes) + 2, nchunks), off) if staged else None
    free_slots = list(range(len(ring))) if staged else None
    fills = []  # every copier future of the pass that may still be running

    def plan(k):
        """chunk k's plan with its host copies as views of the sources"""
        p = dict(plans[k])
        for f in feeds.values():
            p[f.key] = [_Copy(slot, f.src.view(start, cnt, step)) for slot, start, cnt, step in p.get(f.key, ())] if f.host else []
        return p

    def start_fill(p):
        if not staged:
            return
        with N.trace_range("vqa:gather chunk=%d", p["k"]):
            slot = free_slots.pop()
            blk = ring[slot]
            p["slot"], p["ring"], futs = slot, {}, []
            for f in feeds.values():
                if f.staged and f.slots:
                    region = blk[f.ring_off:f.ring_off + f.slots * f.fb].reshape(f.slots, f.fb)
                    p["ring"][f.name] = region
                    futs += _fill_slot(st.copiers(), region, p[f.key])
            p["fill"] = futs
            fills.extend(futs)

    def upload(p):
        """the chunk's frames onto the copy lane, behind the previous chunk's (buffer set k mod nb: the chunk that used it
        last has been waited for)"""
        p["dev"] = {}
        if not host:
            return
        bs = p["k"] % nb
        if p.get("fill"):
            with N.trace_range("vqa:gather-wait chunk=%d", p["k"]):
                for f in p["fill"]:
                    f.result()
                fills[:] = [f for f in fills if not f.done()]
        with N.trace_range("vqa:upload chunk=%d set=%d", p["k"], bs):
            for f in feeds.values():
                if f.host and f.slots and p[f.key]:
                    p["dev"][f.name] = st.device_buffer(bs, f.name, f.slots * f.fb)
                    _upload(cp, p["dev"][f.name].ptr, f.fb, p[f.key], p.get("ring", {}).get(f.name))

    def submit(p, eng):
        ln = p["k"] % len(lanes)
        dev = p["dev"]
        p["has_q"] = p["has_c"] = False
        with N.trace_range("vqa:submit chunk=%d lane=%d", p["k"], ln):
            if host:
                eng.wait_for(cp)   # on the device: the lane's stream continues when the uploads enqueued so far are done
            if want_q:
                fq, fr = feeds["qdist" if split else "dist"], feeds["ref"]
                pair = []
                for f, slot0 in ((fr, 0), (fq, p["qslot"])):
                    if f.host:
                        b = dev[f.name]
                        pair.append(DeviceFrames(b.ptr + slot0 * f.fb, p["qn"], 1, f.fb, frame_stride=f.fb, row_stride=f.fb,
                                                 owner=b, channels=1))
                    else:
                        pair.append(f.src.frames.slice(p["q0"], p["q0"] + p["qn"]))
                eng.quality_submit(pair[0], pair[1], quality.planes, quality.ssim_mode)
                p["has_q"] = True
            if want_c and p["j1"] > p["j0"]:
                m = p["j1"] - p["j0"]
                fd = feeds["dist"]
                if fd.host:
                    h, w = dist.shape[1], dist.shape[2]
                    dd, fb = dev["dist"], fd.fb
                    batch = DeviceFrames(dd.ptr + (1 + p["first"]) * fb, m, h, w, frame_stride=fb * p["step"], owner=dd)
                    prev0 = DeviceFrames(dd.ptr + (1 + p["prev_slot"]) * fb, 1, h, w, owner=dd)
                else:
                    # every frame_interval-th frame, zero-copy: the batch is a strided view of the resident clip
                    s0 = int(idx[1 + p["j0"]])
                    batch = DeviceFrames(dist.ptr + s0 * dist.frame_stride, m, dist.h, dist.w,
                                         frame_stride=dist.frame_stride * complexity.interval, row_stride=dist.row_stride,
                                         owner=dist, channels=dist.channels)
                    prev0 = dist.frame(int(idx[p["j0"]]))
                eng.complexity_submit(batch, prev0, complexity.mask, params)
                p["has_c"] = True

    sse, ssim = [], []
    arrays = {k: [] for k in KINDS + ("temporal",)}  # the same series as float64 arrays, for the pooling (no list round trip)

    def wait(p, eng):
        """block until the chunk is done on its lane; its records are kept, its ring slot is free again"""
        with N.trace_range("vqa:wait chunk=%d lane=%d", p["k"], p["k"] % len(lanes)):
            if p["has_q"]:
                p["qres"] = eng.quality_wait()
            if p["has_c"]:
                p["rec"] = eng.complexity_wait()
        if staged:
            free_slots.append(p["slot"])  # every copy out of the slot has completed

    def finish(p):
        """the host work of a finished chunk: float tails and stats lines (runs AFTER the lane has its next chunk)"""
        with N.trace_range("vqa:tails chunk=%d", p["k"]):
            if p["has_q"]:
                res = p.pop("qres")
                sse.append(res["sse"])
                ssim.append(res["ssim"])
                if on_quality is not None:
                    on_quality(p["q0"], sse[-1], ssim[-1])
            if p["has_c"]:
                rec = p.pop("rec")
                for kind in KINDS:
                    if complexity.mask & MASK[kind]:
                        v = tails.values(kind, rec, complexity.motion_mode)
                        series[kind].extend(tails.as_list(kind, v))
                        arrays[kind].append(v)
                if complexity.mask & N.M_TEMPORAL_DCT:
                    # the reference's first pair only primes prev_gray_frame (:533-537)
                    v = tails.values("temporal", rec)[1 if p["j0"] == 0 else 0:]
                    series["temporal"].extend(tails.as_list("temporal", v))
                    arrays["temporal"].append(v)

    # The loop.  Uploads run one chunk ahead of the kernels and gathers one chunk ahead of the uploads: while the lanes hold
    # chunks k-L+1 .. k, chunk k+1 crosses PCIe (copy lane) and the copiers gather chunk k+2 into the ring.  A finished
    # chunk's host work (tails, stats lines) runs last in an iteration, when the device has everything it can have.
    pending = []
    try:
        cur = plan(0)
        start_fill(cur)
        upload(cur)
        ahead = None
        if nchunks > 1:
            ahead = plan(1)
            start_fill(ahead)
        for k in range(nchunks):
            p, eng = cur, lanes[k % len(lanes)]
            done = None
            if len(pending) == len(lanes):
                done = pending.pop(0)[0]
                wait(done, eng)          # (the oldest pending chunk ran on this very lane)
            submit(p, eng)
            pending.append((p, eng))
            if k + 1 < nchunks:
                cur = ahead
                upload(cur)
                if k + 2 < nchunks:
                    ahead = plan(k + 2)
                    start_fill(ahead)
            if done is not None:
                finish(done)
        while pending:
            p, eng = pending.pop(0)
            wait(p, eng)
            finish(p)
    except BaseException:
        for f in fills:  # a copier may still be writing into the ring: the next pass (or ring()'s free) must not meet it
            try:
                f.result()
            except BaseException:
                pass
        _abandon(lanes + ([cp] if cp is not None else []))
        raise
    q = (np.concatenate(sse), np.concatenate(ssim)) if want_q else None
    if want_c:
        series["_float64"] = {k: (np.concatenate(v).astype(np.float64) if v else np.zeros(0)) for k, v in arrays.items()}
    return q, series


def _abandon(lanes):
    """After a failure inside a pass (the reference's convention: log, re-raise, nothing left running,
    video_processing.py:295-297): wait out whatever the lanes still have pending, so that nothing reads the ring, the
    lane buffers or the caller's frames any more when the error surfaces, and the engines stay usable."""
    for eng in lanes:
        eng.drain()


def _device_of(frames):
    """the device a resident clip lives on (None = the default engine's)"""
    if not isinstance(frames, DeviceFrames):
        return None
    own = getattr(frames, "_owner", None)
    while isinstance(own, DeviceFrames):
        own = getattr(own, "_owner", None)
    if hasattr(own, "engine"):            # a DeviceBuffer knows its engine
        return own.engine.device
    return getattr(getattr(own, "device", None), "index", None)  # a torch tensor its device index
