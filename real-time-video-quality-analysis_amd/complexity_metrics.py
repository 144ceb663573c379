"""Host-side mirror of the reference's complexity interface, backed by the HIP engine.

Same names, argument meaning, return order and error behaviour as
/root/reference/complexity_metrics.py for the hot path:

    calculate_average_scene_complexity   (:246-310)
    process_in_batches                   (:128-148)
    process_frame_complexity / process_dct_frame / process_histogram_frame /
    process_color_histogram_frame / process_edge_frame /
    process_temporal_dct_frame           (:313-579)
    smooth_data, normalize, process_frame_interval_for_parallel,
    calculate_scene_complexity_score     (:114-242)

Differences that are deliberate:
  * frames come from arrays / .npy files / iterables, not cv2.VideoCapture (decode
    is out of scope; no codec exists in the target image);
  * the dispatcher does not fork (a HIP context does not survive fork): a batch of
    frames is one kernel launch instead of one pickled frame per pool task;
  * motion is block-SAD by default (BASELINE.json north_star); set_motion_mode("farneback") or
    VQA_MOTION=farneback selects the reference's own Farneback flow;
  * a clip is read ONCE (stream.py): one fused pass serves all seven series, where the reference decodes
    three times and maps seven times; host clips travel through pinned memory (the caller's, or a ring).
No metric is ever computed on the CPU here; without the HIP library this module
raises on first use.  Like the engines it drives, the module is per process and NOT thread-safe.
"""
import functools
import logging
import os

import numpy as np

from . import _native as N
from . import stream, tails
from .engine import DeviceFrames
from .pooling import pooling_weights
from .pooling import smooth_data as _ewm
from .stream import get_engine, get_engine_pair, release_buffers, selected_indices  # noqa: F401 (re-exported)

logger = logging.getLogger(__name__)


# ---------------------------------------------------------------------------
# frame sources (replaces validate_video_path / read_frame_pairs / extract_frame_timestamps)
# ---------------------------------------------------------------------------
RAW_BGR_SUFFIXES = (".bgr", ".bgr24")


def validate_video_path(input_path):
    """complexity_metrics.py:25-35 with this build's container formats."""
    if not isinstance(input_path, str):
        raise ValueError("Invalid input path. Please provide a valid file path.")
    if input_path.endswith((".npy",) + RAW_BGR_SUFFIXES):
        return "video"
    raise ValueError("Unsupported file type. Please provide a .npy frame stack [N,H,W,3] (uint8, BGR) or a raw .bgr24 stream "
                     "with height and width.")


def _from_torch(t):
    """A PyTorch-ROCm tensor [N,H,W,3] uint8 (north_star: "decoded frames in pinned buffers (PyTorch-ROCm tensors ...)"):
    on the GPU it is used in place (DeviceFrames.from_torch, zero-copy); on the host its memory is handed over as a NumPy
    view - page-locked if the tensor was made with pin_memory=True, which the pass then DMAs from directly."""
    if t.dtype.__str__() != "torch.uint8":
        raise ValueError("frames must be uint8 (got %s)" % t.dtype)
    if t.is_cuda:
        return DeviceFrames.from_torch(t if t.is_contiguous() else t.contiguous())
    return t.numpy()


def _open_frames(video, height=None, width=None):
    """-> array-like [N,H,W,3] uint8 (np.ndarray, memmap or DeviceFrames).  height / width: the geometry of a raw .bgr24 file."""
    if isinstance(video, DeviceFrames):
        return video
    if type(video).__module__.startswith("torch") and hasattr(video, "is_cuda"):
        video = _from_torch(video)
        if isinstance(video, DeviceFrames):
            return video
    if isinstance(video, str):
        validate_video_path(video)
        if not os.path.isfile(video):
            logger.error("Error opening video file: %s", video)
            return np.zeros((0, 1, 1, 3), np.uint8)
        if video.endswith(RAW_BGR_SUFFIXES):
            if not height or not width:
                raise ValueError("a raw BGR24 stream needs height and width (config keys or arguments of the same names)")
            from .frames import open_raw_bgr24
            return open_raw_bgr24(video, height, width)
        return np.load(video, mmap_mode="r")
    arr = video if isinstance(video, np.ndarray) else np.stack([np.asarray(f) for f in video])
    if arr.ndim != 4 or arr.shape[3] != 3 or arr.dtype != np.uint8:
        raise ValueError("frames must be uint8 [N,H,W,3] packed BGR")
    return arr


def _num_frames(fr):
    return fr.n if isinstance(fr, DeviceFrames) else fr.shape[0]


def read_frame_pairs(video_path, frame_interval=10):
    """complexity_metrics.py:76-111 — list of (current, previous) selected frames."""
    fr = _open_frames(video_path)
    if isinstance(fr, DeviceFrames):
        raise TypeError("read_frame_pairs returns host frames; pass host frames")
    idx = selected_indices(fr.shape[0], frame_interval)
    return [(np.asarray(fr[idx[j]]), np.asarray(fr[idx[j - 1]])) for j in range(1, len(idx))]


def extract_frame_timestamps(video_path, frame_interval=10, fps=30.0):
    """complexity_metrics.py:38-73 for a constant-rate source: timestamps (ms) of the
    frames whose 0-based index % interval == 0 (:65; note the different phase from
    read_frame_pairs).  OpenCV's CAP_PROP_POS_MSEC after reading frame i is i*1000/fps."""
    fr = _open_frames(video_path)
    n = _num_frames(fr)
    return [float(i) * 1000.0 / fps for i in range(0, n, frame_interval)]


# ---------------------------------------------------------------------------
# pooling
# ---------------------------------------------------------------------------
def smooth_data(data, alpha=0.8):
    """complexity_metrics.py:114-125 (CPU branch): pandas ewm(alpha).mean() as a NumPy array."""
    return _ewm(data, alpha)


def normalize(value, min_value, max_value):
    """complexity_metrics.py:167-169."""
    return (value - min_value) / (max_value - min_value) if max_value > min_value else 0


def process_frame_interval_for_parallel(timestamps):
    """complexity_metrics.py:150-165 — frame rate between two timestamps (ms)."""
    prev_timestamp, curr_timestamp = timestamps
    frame_interval = (curr_timestamp - prev_timestamp) / 1000.0
    if frame_interval > 0:
        return 1.0 / frame_interval
    return 0.0


# ---------------------------------------------------------------------------
# counts -> reference scalars (the float tails the reference runs in NumPy)
# ---------------------------------------------------------------------------
_gray_entropy = tails.gray_entropy      # :413-414
_color_entropy = tails.color_entropy    # :464-473

_MOTION_MODES = {"sad": N.MOTION_SAD, "farneback": N.MOTION_FARNEBACK}
_motion_mode = _MOTION_MODES[os.environ.get("VQA_MOTION", "sad")]


def set_motion_mode(mode):
    """What the motion slot of the 8-tuple / process_frame_complexity computes: "sad" (north_star's 16x16
    block-SAD mean |mv|, the default) or "farneback" (cv2.calcOpticalFlowFarneback mean magnitude, :340-343)."""
    global _motion_mode
    _motion_mode = _MOTION_MODES[mode]


def _scalar(kind, rec):
    return tails.scalar(kind, rec, _motion_mode)


def _scalars(kind, rec):
    """the batch form: same values, same types, one pass over the records (tails.py)"""
    return tails.scalars(kind, rec, _motion_mode)


_MASK = stream.MASK


# ---------------------------------------------------------------------------
# per-frame callables (same signatures as the reference's process_* functions)
# ---------------------------------------------------------------------------
def _one(kind, frame, resize_width, resize_height):
    frame = np.asarray(frame)
    if frame.ndim == 2:  # gray input: B=G=R=v makes cvtColor the identity (3735+19235+9798 = 2^15)
        frame = np.repeat(frame[..., None], 3, axis=2)
    eng = get_engine()
    with stream.pass_lock(eng.device):
        rec = eng.complexity(frame[None], mask=_MASK[kind], resize=(resize_width, resize_height))
    return _scalar(kind, rec[0])


def process_dct_frame(frame, resize_width, resize_height):
    """complexity_metrics.py:346-364 — sum of squared DCT coefficients of resize(gray(frame))."""
    return _one("dct", frame, resize_width, resize_height)


def process_histogram_frame(frame, resize_width, resize_height):
    """complexity_metrics.py:392-416 — Shannon entropy of the gray histogram."""
    return _one("hist", frame, resize_width, resize_height)


def process_color_histogram_frame(frame, resize_width, resize_height):
    """complexity_metrics.py:418-475 — summed entropy of the B, G, R histograms."""
    return _one("color", frame, resize_width, resize_height)


def process_edge_frame(frame, resize_width, resize_height):
    """complexity_metrics.py:477-504 — number of Canny(100,200) edge pixels."""
    return _one("edge", frame, resize_width, resize_height)


def process_frame_complexity(frame_pair):
    """complexity_metrics.py:313-343 — motion between (current, previous): block-SAD mean |mv|, or the
    Farneback mean flow magnitude after set_motion_mode("farneback")."""
    frame, prev_frame = frame_pair
    if frame is None or prev_frame is None:
        return 0.0
    eng = get_engine()
    with stream.pass_lock(eng.device):
        rec = eng.complexity(np.asarray(frame)[None], prev0=np.asarray(prev_frame), mask=N.M_MOTION,
                             motion_mode=_motion_mode)
    return _scalar("motion", rec[0])


def process_temporal_dct_frame(prev_gray_frame, curr_gray_frame, resize_width, resize_height):
    """complexity_metrics.py:543-579 — sum |dct(prev) - dct(curr)| of two gray frames."""
    prev3 = np.repeat(np.asarray(prev_gray_frame)[..., None], 3, axis=2)
    curr3 = np.repeat(np.asarray(curr_gray_frame)[..., None], 3, axis=2)
    eng = get_engine()
    with stream.pass_lock(eng.device):
        rec = eng.complexity(curr3[None], prev0=prev3, mask=N.M_TEMPORAL_DCT, resize=(resize_width, resize_height))
    return _scalar("temporal", rec[0])


def process_orb_frame_for_parallel(frame):
    """complexity_metrics.py:367-389 — number of ORB keypoints of the 64x64 thumbnail (the reference
    hard-codes the size and every ORB parameter at its default)."""
    frame = np.asarray(frame)
    if frame.ndim == 2:
        frame = np.repeat(frame[..., None], 3, axis=2)
    eng = get_engine()
    with stream.pass_lock(eng.device):
        rec = eng.complexity(frame[None], mask=N.M_ORB)
    return _scalar("orb", rec[0])


_KNOWN = {process_dct_frame: "dct", process_histogram_frame: "hist", process_color_histogram_frame: "color",
          process_edge_frame: "edge", process_frame_complexity: "motion", process_orb_frame_for_parallel: "orb"}


def _unwrap(func, kwargs):
    """Peel functools.partial layers; returns (base function, merged keyword arguments)."""
    kw = dict(kwargs)
    args = ()
    while isinstance(func, functools.partial):
        merged = dict(func.keywords or {})
        merged.update(kw)
        kw = merged
        args = tuple(func.args) + args
        func = func.func
    return func, args, kw


def process_in_batches(frames, process_func, num_workers, batch_size=100, **kwargs):
    """complexity_metrics.py:128-148 — ordered map of process_func over frames in chunks.

    When process_func is one of this module's kernels the whole chunk becomes ONE
    batched launch on the device (num_workers is accepted and ignored: there is no
    process pool to size).  Any other callable is mapped in-process, in order."""
    base, pargs, kw = _unwrap(process_func, kwargs)
    kind = _KNOWN.get(base)
    results = []
    if kind is None or pargs:
        call = functools.partial(process_func, **kwargs)
        for i in range(0, len(frames), batch_size):
            results.extend(call(item) for item in frames[i:i + batch_size])
        return results
    eng = get_engine()
    with stream.pass_lock(eng.device):  # the ring, the default engine's pending state: one pass at a time per device
        return _batched_map(eng, kind, frames, process_func, batch_size, kw, kwargs)


def _batched_map(eng, kind, frames, process_func, batch_size, kw, kwargs):
    """process_in_batches' kernel branch: every chunk is ONE pinned buffer (gathered by the copier threads), ONE asynchronous
    upload, ONE launch, in the pipeline of stream.run: uploads go through the device's copy lane (in chunk order), chunks
    alternate between two measuring engines that wait for their upload on the device - while chunk k runs, chunk k + 1
    crosses PCIe and the copiers gather chunk k + 2 into the ring."""
    resize = None if kind in ("orb", "motion") else (kw["resize_width"], kw["resize_height"])
    params = eng.make_params(resize=resize, motion_mode=_motion_mode)
    call = functools.partial(process_func, **kwargs)
    chunks = [_chunk_job(kind, frames[i:i + batch_size], call) for i in range(0, len(frames), batch_size)]
    out = [None] * len(chunks)
    # chunks that cannot be one launch (frames of different sizes: per-item map; a chunk of None pairs: zeros) first, while
    # the default engine is idle; results keep their place
    for k, c in enumerate(chunks):
        if not c["items"]:
            out[k] = c["finish"](None)
    order = [k for k, c in enumerate(chunks) if c["items"]]
    if not order:
        return [v for part in out for v in part]
    lanes = list(stream.get_engine_lanes(eng.device, 2)) if len(order) > 1 and stream.MAX_LANES > 1 else [eng]
    cp = stream.get_copy_engine(eng.device)
    st = stream._staging_of(eng)
    nb, nslots = len(lanes) + 1, min(len(lanes) + 2, len(order))
    # ring slots and buffer sets are sized ONCE for the largest chunk (a slot is never re-allocated while another is in flight)
    slot_bytes = max(sum(int(f.nbytes) for f in chunks[k]["items"]) for k in order)
    fills = []      # copier futures that may still be writing into the ring
    gathered = {}

    def gather(i):
        arr, futs = stream.stage_frames_start(eng, chunks[order[i]]["items"], slot=i % nslots, slots=nslots, min_bytes=slot_bytes)
        fills.extend(futs)
        gathered[i] = (arr, futs)

    def upload(i):
        arr, futs = gathered.pop(i)
        for f in futs:
            f.result()
        fills[:] = [f for f in fills if not f.done()]
        buf = st.device_buffer(i % nb, "pib", slot_bytes)
        cp.h2d_async(buf.ptr, arr.ctypes.data, arr.nbytes)
        return buf, arr.shape

    def submit(i, up):
        lane = lanes[i % len(lanes)]
        buf, (n, h, w, _c) = up
        lane.wait_for(cp)   # on the device: the lane continues when the uploads enqueued so far have landed
        fr = DeviceFrames(buf.ptr, n, h, w, owner=buf)
        if chunks[order[i]]["prev0"]:
            lane.complexity_submit(fr.slice(1, n), fr.frame(0), _MASK[kind], params)
        else:
            lane.complexity_submit(fr, None, _MASK[kind], params)

    def wait(i):
        k = order[i]
        out[k] = chunks[k]["finish"](lanes[i % len(lanes)].complexity_wait())

    try:
        gather(0)
        up = upload(0)
        if len(order) > 1:
            gather(1)
        for i in range(len(order)):
            if i >= len(lanes):
                wait(i - len(lanes))    # frees the lane, the buffer set chunk i + 1 goes into and the ring slot of chunk i + 2
            submit(i, up)
            if i + 1 < len(order):
                up = upload(i + 1)
                if i + 2 < len(order):
                    gather(i + 2)
        for i in range(max(len(order) - len(lanes), 0), len(order)):
            wait(i)
    except BaseException:
        for f in fills:  # (every copier of this call: the ring must be quiet when the error surfaces)
            try:
                f.result()
            except BaseException:
                pass
        for lane in lanes + [cp]:
            lane.drain()
        raise
    return [v for part in out for v in part]


def _chunk_job(kind, batch, call):
    """One chunk of process_in_batches as a launch: the frames to stage (items; None = nothing to launch), whether slot 0 is
    the frame before the first (prev0), and finish(records) -> the chunk's results in item order."""
    if kind != "motion":
        # 2-D (gray) frames are expanded exactly as the per-frame callables do; a chunk whose frames differ in size
        # cannot be one launch and goes through the per-item map (what the reference's executor.map does anyway)
        items = [np.asarray(f) for f in batch]
        items = [np.repeat(f[..., None], 3, axis=2) if f.ndim == 2 else f for f in items]
        if len({f.shape for f in items}) > 1:
            return dict(items=None, prev0=False, finish=lambda rec: [call(item) for item in batch])
        return dict(items=items, prev0=False, finish=lambda rec: _scalars(kind, rec))
    pairs = batch
    live = [j for j, p in enumerate(pairs) if p[0] is not None and p[1] is not None]

    def fill(values):
        out = [0.0] * len(pairs)  # a pair with a None frame is 0.0 (:324-325)
        for j, v in zip(live, values):
            out[j] = v
        return out

    if not live:
        return dict(items=None, prev0=False, finish=lambda rec: fill([]))
    chained = all(pairs[live[k]][1] is pairs[live[k - 1]][0] for k in range(1, len(live))) and \
        live == list(range(live[0], live[0] + len(live)))
    if chained:
        # (previous of the first pair, then every current frame) in one pinned buffer: slot 0 is the halo
        items = [np.asarray(pairs[live[0]][1])] + [np.asarray(pairs[j][0]) for j in live]
        return dict(items=items, prev0=True, finish=lambda rec: fill(_scalars("motion", rec)))
    # arbitrary pairs: interleave (prev, curr) and keep every second result
    items = [np.asarray(pairs[j][s]) for j in live for s in (1, 0)]
    return dict(items=items, prev0=False, finish=lambda rec: fill(_scalars("motion", rec[1::2])))


# ---------------------------------------------------------------------------
# the aggregator
# ---------------------------------------------------------------------------
def complexity_series(video, resize_width, resize_height, frame_interval=10, batch_size=100, engine=None,
                      dct_mode=N.DCT_AUTO, mask=N.M_ALL, shard=None, motion=None, device=None):
    """Per-frame series for the frames the reference measures, from ONE fused pass (stream.run).

    Returns dict kind -> list, each in the reference's sample order:
      motion/dct/hist/edge/orb/color : T-1 samples (selected frames S_1..S_{T-1}; :268-290)
      temporal                   : T-2 samples (S_1->S_2 ... ; :533-537)
    shard=(rank, world): only this rank's contiguous range of the T-1 samples is computed (the
    shard also reads the ONE selected frame before its first, as the pair metrics' halo); the lists then hold
    that range only and out["range"] = (lo, hi) gives its place in the whole series.
    motion: "sad" / "farneback" (None: set_motion_mode's / VQA_MOTION's choice); device: the GPU of a host clip's pass
    (None: VQA_DEVICE, else LOCAL_RANK, else 0) - the config keys of the same names.
    Chunks of up to batch_size samples alternate between two engines of the device; only the selected frames of a host clip
    cross PCIe (through the copy lane), from the caller's pinned memory or through the pinned ring."""
    cx = stream.Complexity((resize_width, resize_height), frame_interval, mask, dct_mode, motion_mode_of(motion), shard)
    return stream.run(_open_frames(video), complexity=cx, batch_size=batch_size, engine=engine, device=device)[1]


def motion_mode_of(motion=None):
    """"sad" / "farneback" / None (the module's current mode) -> the C ABI's constant"""
    if motion is None:
        return _motion_mode
    if motion not in _MOTION_MODES:
        raise ValueError("motion must be 'sad' or 'farneback'.")
    return _MOTION_MODES[motion]


_DCT_MODES = {None: N.DCT_AUTO, "auto": N.DCT_AUTO, "full": N.DCT_FULL, "block8": N.DCT_BLOCK8}


def calculate_temporal_dct(video_path, resize_width, resize_height, frame_interval=10, smoothing_factor=0.8,
                           dct_mode=None):
    """complexity_metrics.py:506-541.  dct_mode: None/"auto" (full-frame up to 128x128, 8x8 blocks above),
    "full" (the reference's own full-frame cv2.dct metric at any size) or "block8" (north_star's metric)."""
    s = complexity_series(video_path, resize_width, resize_height, frame_interval, mask=N.M_TEMPORAL_DCT,
                          dct_mode=_DCT_MODES[dct_mode])["temporal"]
    sm = smooth_data(s, smoothing_factor)
    return np.mean(sm) if len(sm) > 0 else 0.0


def pool_series(s, video_path, frame_interval=10, smoothing_factor=0.8, num_workers=None, batch_size=100, fps=30.0):
    """The pooling half of calculate_average_scene_complexity (:297-310): EWM-smooth and mean every series, the
    frame-rate variation from the timestamps; returns the 8-tuple in the reference's order."""
    def pooled(x):
        with np.errstate(invalid="ignore"), _quiet_empty_mean():
            return np.mean(smooth_data(x, smoothing_factor))

    # the pass keeps every series as one float64 array too (the values the lists hold, without 256 NumPy scalars per list)
    s = dict(s, **s.get("_float64", {}))
    temporal = smooth_data(s["temporal"], smoothing_factor)
    temporal_dct_complexity = np.mean(temporal) if len(temporal) > 0 else 0.0
    frame_timestamps = extract_frame_timestamps(video_path, frame_interval, fps)
    timestamp_pairs = list(zip(frame_timestamps[:-1], frame_timestamps[1:]))
    framerate_variation = process_in_batches(timestamp_pairs, process_frame_interval_for_parallel, num_workers, batch_size)
    return (pooled(s["motion"]), pooled(s["dct"]), pooled(s["hist"]), pooled(s["edge"]), pooled(s["orb"]),
            pooled(s["color"]), temporal_dct_complexity, pooled(framerate_variation))


def calculate_average_scene_complexity(video_path, resize_width, resize_height, frame_interval=10,
                                       smoothing_factor=0.8, num_workers=None, batch_size=100, fps=30.0,
                                       dct_mode=None, motion=None, device=None):
    """complexity_metrics.py:246-310.  Returns the 8-tuple in the reference's order (:301-310):
    (motion, dct, histogram, edge, orb, colour_histogram, temporal_dct, framerate_variation).
    `video_path` may be a .npy path, an ndarray [N,H,W,3] or DeviceFrames; `fps` stands in for the container's
    timestamps (:66); `dct_mode` is as for calculate_temporal_dct, `motion` / `device` as for complexity_series."""
    if dct_mode not in _DCT_MODES:
        raise ValueError("dct_mode must be 'auto', 'block8' or 'full'.")
    s = complexity_series(video_path, resize_width, resize_height, frame_interval, batch_size,
                          dct_mode=_DCT_MODES[dct_mode], motion=motion, device=device)
    return pool_series(s, video_path, frame_interval, smoothing_factor, num_workers, batch_size, fps)


def calculate_average_scene_complexity_sharded(video_path, resize_width, resize_height, frame_interval=10,
                                               smoothing_factor=0.8, batch_size=100, fps=30.0, group=None,
                                               series_fn=None):
    """calculate_average_scene_complexity of ONE long stream on N GPUs, one process per GPU.

    Every rank measures a contiguous range of the selected frames (plus the one frame before it) on its own
    device, no frame ever moves between GPUs.  mean(EWM(x)) is a fixed linear functional sum_i c_i x_i
    (pooling.pooling_weights), so each rank forms its partial sums with GLOBAL sample indices and ONE float64
    SUM all-reduce of 7 scalars (RCCL when the process group is "nccl") yields, on every rank, the 8-tuple the
    single-process function returns (to ~1e-15 relative: the reduction order differs).
    Requires an initialised torch.distributed process group; `video_path` as for the single-process call.
    `series_fn` replaces complexity_series (tests inject a CPU stand-in; the product default is the HIP path)."""
    import torch
    import torch.distributed as td
    rank, world = td.get_rank(group), td.get_world_size(group)
    s = (series_fn or complexity_series)(video_path, resize_width, resize_height, frame_interval, batch_size,
                                         shard=(rank, world))
    lo, hi = s["range"]
    T = len(selected_indices(_num_frames(_open_frames(video_path)), frame_interval)) - 1  # samples in the whole series
    kinds = ("motion", "dct", "hist", "edge", "orb", "color")
    part = np.zeros(len(kinds) + 1, np.float64)
    if T > 0 and hi > lo:
        c = pooling_weights(T, smoothing_factor)[lo:hi]
        for k, kind in enumerate(kinds):
            part[k] = float(np.dot(c, np.asarray(s[kind], np.float64)))
        if T > 1:  # temporal sample j (global) belongs to per-frame sample j + 1
            tl = max(lo, 1) - 1
            ct = pooling_weights(T - 1, smoothing_factor)[tl:tl + len(s["temporal"])]
            part[-1] = float(np.dot(ct, np.asarray(s["temporal"], np.float64)))
    backend = td.get_backend(group)
    t = torch.from_numpy(part)
    if backend == "nccl":  # RCCL reduces device tensors: use this rank's engine device, not torch's default
        t = t.to("cuda:%d" % get_engine().device)
    td.all_reduce(t, op=td.ReduceOp.SUM, group=group)
    tot = t.cpu().numpy()
    nan = float("nan")
    vals = [float(v) if T > 0 else nan for v in tot[:len(kinds)]]
    temporal = float(tot[-1]) if T > 1 else 0.0  # :541
    frame_timestamps = extract_frame_timestamps(video_path, frame_interval, fps)
    timestamp_pairs = list(zip(frame_timestamps[:-1], frame_timestamps[1:]))
    fv = [process_frame_interval_for_parallel(p) for p in timestamp_pairs]
    with np.errstate(invalid="ignore"), _quiet_empty_mean():
        fpsv = np.mean(smooth_data(fv, smoothing_factor))
    return (vals[0], vals[1], vals[2], vals[3], vals[4], vals[5], temporal, fpsv)


class _quiet_empty_mean:
    """np.mean([]) warns and returns NaN — the reference's behaviour for an unopenable video (:56-58)."""

    def __enter__(self):
        import warnings
        self._cm = warnings.catch_warnings()
        self._cm.__enter__()
        warnings.simplefilter("ignore", RuntimeWarning)

    def __exit__(self, *a):
        return self._cm.__exit__(*a)


def calculate_scene_complexity_score(encoded_video, resize_width, resize_height, frame_interval=10,
                                     smoothing_factor=0.8, num_workers=None, batch_size=100):
    """complexity_metrics.py:171-242 — min/max normalisation (:197-206) and weights (:219-228)."""
    (motion, dct, hist, edge, orb, color, temporal, fpsvar) = calculate_average_scene_complexity(
        encoded_video, resize_width, resize_height, frame_interval=frame_interval,
        smoothing_factor=smoothing_factor, num_workers=None, batch_size=batch_size)
    table = (  # (value, min, max, weight)
        (motion, 0.0, 10.0, 0.25), (dct, 1e6, 5e7, 0.15), (temporal, 0.0, 1e7, 0.15), (hist, 0.0, 8.0, 0.10),
        (edge, 0.0, 1.0, 0.10), (orb, 0.0, 5000, 0.10), (color, 0.0, 8.0, 0.10), (fpsvar, 0.0, 2.0, 0.05))
    return sum(normalize(v, lo, hi) * wt for v, lo, hi, wt in table)
