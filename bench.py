#!/usr/bin/env python3
"""bench.py — frames/sec of the hot path on synthetic frame streams, one process per GPU.

  python bench.py --gpus 1 --steps K --warmup W                  (N = 1)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W   (N > 1, one rank per GPU)
  python bench.py --gpus N ...                                   (N > 1 with no launcher: bench.py starts that very
         torch.distributed.run command itself as a child process, relays rank 0's line and the exit code)

A "step" is one pass of the selected workload over one device-resident batch of
frames.  `value` is timed on the product's own configuration: block-SAD and the
Canny chain on the library's side streams (VQA_OPT_OVERLAP, its default) and two
batches in flight, each on its own context (--inflight 2: step i+1 is submitted
before step i is waited for - what a host that streams batches does; measured best
of {1,2} contexts per batch x {1,2,3} batches in flight, LAB_NOTES.md), per-kernel
profiling OFF.  Because overlapped kernels share the GPU, their event times say
nothing about a kernel alone; so after the timed region a SERIAL pass runs the same
steps on one context with the overlap off and HIP-event profiling on: it fills
"kernels" / "roofline" and is reported as "ms_per_step_serial" (never `value`).
Rank 0 prints ONE JSON line (contract in the task statement) with four
extra objects: "roofline" (dominant kernel: algorithmic bytes / HIP-event time
against the HBM peak, or - for a kernel far below that roof that does counted fp32
work - flops against the fp32 vector peak, "bound": "valu_fp32"), "end_to_end"
(the same workload fed from pinned host memory over PCIe, measured after the
timed region; never `value`), at N = 1 "cpu_baseline" (the oracle port under a
process pool that reproduces the reference's process_in_batches, timed on this
box's host cores) and "verified": the result records of the LAST TIMED step at a
few batch positions are compared with what the oracle expects for those frames
(expectations computed before the process touches the GPU); a mismatch is fatal
on every rank (exit code 4) and no line is printed.  A full-suite workload's line
also carries "api_end_to_end": the SAME clip, in the workload's own modes (config keys
ssim_mode / pixfmt / dct_mode / motion), through
the reference-shaped Python entry point process_video_and_extract_metrics
(video_processing.py:216 + :242: run_ffmpeg_metrics + calculate_average_scene_complexity,
one pass) from HBM, from pinned host memory and from pageable host memory - the
drop-in surface on the measured path, stats files and CSV row included; never `value`.

Workloads (BASELINE.json configs):
  c3  1920x1080 full complexity suite + PSNR/SSIM          [default: the config BASELINE.json's metric is quoted on]
  c2  1920x1080, frame_interval=1, PSNR + SSIM (Gaussian) + 8x8 DCT (energy + temporal)
  c4  3840x2160 full suite
  c1  BASELINE configs[0], the reference's own config.json: 64x64 resize, frame_interval=10 on a 300-frame 1080p
      clip, through process_video_and_extract_metrics (PSNR/SSIM of all 300 frame pairs at 1080p + the complexity
      suite of every 10th frame at 64x64, full-frame DCT as the reference computes it at that size).  A step = one
      call on the clip; `value` = source frames/s with the clip resident in HBM, api_end_to_end = from host memory
  c1ref  c1 with the REFERENCE's own definitions through the same entry point: the quality pair as decoded yuv420p planes
      (what FFmpeg's psnr / ssim filters compare, video_processing.py:274-276) with vf_ssim's 8x8 integer windows, the
      encoded stream's BGR frames beside it for the complexity suite (complexity_metrics.py:100) with Farneback motion (:340)
      and the full-frame 64x64 DCT: config keys ssim_mode=ffmpeg, pixfmt=yuv420p, motion=farneback
  c3ref  1920x1080, the REFERENCE's own definitions in one line: Farneback motion (complexity_metrics.py:340),
      full-frame DCT energy + temporal L1 (:363, :574-579), FFmpeg vf_ssim + psnr on yuv420p planes
      (video_processing.py:275-276), gray + colour histograms, Canny, ORB count; 64 frames per step
Frame streams shard one-stream-per-GPU (weak scaling): every rank runs the same
workload on its own stream; the only cross-rank traffic is one scalar
all-reduce (RCCL) of the pooled metrics after the timed region.  An RCCL failure
at N > 1 is fatal on every rank (exit code 3).  VQA_BENCH_DEVICE pins every rank
to one device: the 1-GPU rehearsal of the rank logic, which reduces over gloo.
"""
import argparse
import glob
import hashlib
import json
import os
import re
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured copy)
CSRC = os.path.join(REPO, "real-time-video-quality-analysis_amd", "csrc")

WORKLOADS = {
    "c1": dict(h=1080, w=1920, batch=300, full=True, api=True,
               name="config.json defaults: 300 x 1920x1080 clip, resize 64x64, frame_interval=10, "
                    "process_video_and_extract_metrics = PSNR/SSIM(gauss) of every frame pair + complexity suite of every 10th frame"),
    "c1ref": dict(h=1080, w=1920, batch=300, full=True, api=True, ref_true=True,
                  name="config.json defaults with the REFERENCE's own definitions: 300 x 1920x1080 clip, resize 64x64, frame_interval=10, "
                       "process_video_and_extract_metrics(r.yuv420p, d.yuv420p, encoded_bgr) = FFmpeg psnr + vf_ssim on the decoded Y,U,V planes "
                       "of every frame pair + complexity suite of every 10th BGR frame with Farneback motion and the full-frame 64x64 DCT"),
    "c2": dict(h=1080, w=1920, batch=256, full=False,
               name="1920x1080 frame_interval=1 PSNR+SSIM(gauss 11x11)+8x8 DCT(energy+temporal), BGR24 pairs"),
    "c3": dict(h=1080, w=1920, batch=256, full=True,
               name="1920x1080 full suite (motion-SAD, DCT, temporal-DCT, Canny, ORB count, gray+colour hist) + PSNR/SSIM"),
    "c4": dict(h=2160, w=3840, batch=64, full=True, name="3840x2160 full suite + PSNR/SSIM"),
    "c3ref": dict(h=1080, w=1920, batch=64, full=True,
                  defaults=dict(motion="farneback", dct_mode="full", ssim_mode="ffmpeg", pixfmt="yuv420p", streams=2, inflight=1),
                  name="1920x1080 reference-true suite (Farneback motion, full-frame DCT + temporal DCT, Canny, ORB count, "
                       "gray+colour hist) + FFmpeg psnr/vf_ssim on yuv420p planes"),
}


# ---------------------------------------------------------------------------
# CPU baseline: oracle port, dispatcher = the reference's process_in_batches
# (ProcessPoolExecutor, executor.map chunksize 1 => one pickle per item).
# ---------------------------------------------------------------------------
def _cpu_item(item):
    from oracle import c_oracle as co
    from oracle import pipeline as pl
    from rtvqa_amd.engine import bgr_planes
    ref, dist, prev, full, motion = item
    h, w = dist.shape[:2]
    out = []
    sse, ssim = pl.frame_quality(ref, dist, bgr_planes(h, w), "gauss")
    out += sse + ssim
    g, gp = co.bgr2gray(dist), co.bgr2gray(prev)
    out += list(co.dct8x8(gp, g)[:2])
    if full:
        out.append(co.canny(g, 100, 200)[0])
        out += [co.farneback(gp, g)] if motion == "farneback" else list(co.block_sad(gp, g, 7)[:2])
        out.append(int(co.hist_u8(g).sum()))
        out += [int(co.hist_u8(dist, offset=c, step=3).sum()) for c in range(3)]
        out.append(pl.process_orb_frame_for_parallel(dist))
    return out


def stream_chunk(content, rank, h, w, a, n):
    """Frames a .. a+n-1 of rank's synthetic stream (reference, distorted).  `a` is a multiple of CHUNK for noise
    (its generator is seeded per chunk); the upload loop and the verification regenerate frames through this."""
    from rtvqa_amd import synth
    r = (synth.s_natural(n, h, w, seed=1234, stream_id=rank, t0=a) if content == "natural"
         else synth.s_noise(n, h, w, seed=1234 + a, stream_id=rank))
    return r, synth.distort(r, t0=a)


CHUNK = 32


def verify_frames(B):
    """Batch positions checked against the oracle: both ends, the middle, and both sides of the first 16-frame
    seam (the marching DCT's chunk boundary)."""
    return sorted(set(j for j in (0, 1, 15, 16, 17, B // 2, B - 1) if 0 <= j < B))


def _expect_item(item):
    """Oracle's expected records for batch position j of the stream (stream index j+1; previous = stream index j)."""
    from oracle import check
    content, rank, h, w, B, j, full, motion, ssim_mode, yuv, dct_mode = item
    if content == "natural":
        r, d = stream_chunk(content, rank, h, w, j, 2)
        ref, dist, prev = r[1], d[1], d[0]
    else:  # noise: regenerate the chunk(s) holding stream frames j and j+1
        got = {}
        for i in (j, j + 1):
            a = i // CHUNK * CHUNK
            r, d = stream_chunk(content, rank, h, w, a, min(CHUNK, B + 1 - a))
            got[i] = (r[i - a].copy(), d[i - a].copy())
        ref, dist, prev = got[j + 1][0], got[j + 1][1], got[j][1]
    planes = qpair = None
    if yuv:
        from rtvqa_amd.engine import yuv420p_planes
        from rtvqa_amd.frames import bgr_to_yuv420p
        planes = yuv420p_planes(h, w)
        qpair = (bgr_to_yuv420p(ref[None])[0], bgr_to_yuv420p(dist[None])[0])
    return j, check.expected(ref, dist, prev, full, (ssim_mode,), motion, planes, qpair, dct_mode)


def visible_cores():
    try:
        return len(os.sched_getaffinity(0))  # the cores this process may run on
    except AttributeError:
        return os.cpu_count() or 1


def cgroup_cpu_limit():
    """CPU time the container may use, in cores (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited/unknown.
    Informational: the reference sizes its pool from os.cpu_count() and never looks at this."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else round(int(q) / int(p), 2)
    except Exception:
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else round(q / p, 2)
    except Exception:
        return None


def cpu_workers(cores):
    """num_workers exactly as the reference picks it (complexity_metrics.py:264-265): cores // 2.
    VQA_CPU_WORKERS overrides it explicitly (reported as workers_override)."""
    ov = os.environ.get("VQA_CPU_WORKERS")
    if ov:
        return max(1, int(ov)), True
    return max(1, cores // 2), False


def cpu_baseline(h, w, full, sample, motion="sad", verify_items=()):
    from concurrent.futures import ProcessPoolExecutor
    from oracle import c_oracle as co
    from rtvqa_amd import synth
    co.build()
    cores = visible_cores()
    workers, override = cpu_workers(cores)
    if sample <= 0:  # bounded sample: every worker gets a few items, ~10-30 s of CPU work (scaled with the frame area)
        sample = max(8, int(max(32, 3 * workers) * (1920.0 * 1080.0) / (h * w)))
    ref = synth.s_natural(sample, h, w, seed=1234, stream_id=0, t0=0)
    dist = synth.distort(ref, t0=0)
    items = [(ref[i], dist[i], dist[i - 1] if i else dist[0], full, motion) for i in range(sample)]
    t0 = time.perf_counter()
    results = []
    with ProcessPoolExecutor(max_workers=workers) as ex:
        for i in range(0, len(items), 100):  # batch_size=100, barrier per batch (:144-147)
            results.extend(ex.map(_cpu_item, items[i:i + 100]))
        dt = time.perf_counter() - t0
        expect = dict(ex.map(_expect_item, verify_items))  # the checker for "verified": same pool, not timed
    # `cores` = the worker processes that actually ran (the contract's "threads you actually used"); visible_cores = what
    # os.sched_getaffinity shows the process, which is what the reference's cpu_count() // 2 rule starts from
    line = dict(value=round(sample / dt, 3), unit="frames/s", cores=workers, visible_cores=cores, workers=workers, workers_override=override,
                cgroup_cpu_limit=cgroup_cpu_limit(), kind="port", seconds=round(dt, 2),
                sample="%d frame pairs of the same workload, oracle/ C port under ProcessPoolExecutor(max_workers="
                       "cores//2 = %d of the %d cores visible to the process), chunksize 1, batch_size 100"
                       % (sample, workers, cores))
    line.update(second_cpu_figure(workers, override, lambda wk, m: _timed_pool(_cpu_item, items[:m], wk), sample))
    return expect, finish_cpu_line(line)


def _timed_pool(fn, items, workers):
    from concurrent.futures import ProcessPoolExecutor
    t0 = time.perf_counter()
    with ProcessPoolExecutor(max_workers=workers) as ex:
        for i in range(0, len(items), 100):
            list(ex.map(fn, items[i:i + 100]))
        return time.perf_counter() - t0


def second_cpu_figure(workers, override, run, sample):
    """The reference sizes its pool from os.cpu_count() and never looks at the container's CPU quota: on a box whose cgroup
    allows fewer cores than it shows, its rule oversubscribes.  The same items with one worker per core the cgroup really
    grants, on half the sample - finish_cpu_line() then reports the FASTER of the two as `value`."""
    lim = cgroup_cpu_limit()
    if override or not lim or workers <= lim:
        return {}
    wk = max(1, int(lim))
    m = max(wk, sample // 2)
    dt = run(wk, m)
    return {"at_cgroup_limit": dict(value=round(m / dt, 3), workers=wk, seconds=round(dt, 2),
                                    sample="%d of the same items, one worker per core the cgroup grants" % m)}


def finish_cpu_line(line):
    """`value` is the honest CPU figure: the faster of (the reference's rule: cores // 2 workers of the cores the process SEES)
    and (one worker per core the cgroup GRANTS).  `cores` = the cores that actually ran it: min(workers, cgroup limit).  The
    other figure stays beside it (reference_rule / at_cgroup_limit), so a ratio taken from `value` is never inflated by
    an oversubscribed pool (128 workers on a 16-core cgroup ran 1.7x slower than 16)."""
    lim = line.get("cgroup_cpu_limit")
    alt = line.get("at_cgroup_limit")
    line["cores"] = int(min(line["workers"], lim)) if lim else line["workers"]
    if alt and alt["value"] > line["value"]:
        line["reference_rule"] = dict(value=line["value"], workers=line["workers"], seconds=line["seconds"],
                                      what="ProcessPoolExecutor(max_workers = visible cores // 2), as complexity_metrics.py:264-265 "
                                           "sizes it: oversubscribes a cgroup that grants fewer cores than it shows")
        line.update(value=alt["value"], seconds=alt["seconds"], workers=alt["workers"], cores=alt["workers"],
                    value_is="at_cgroup_limit (the faster of the two pool sizes)")
        line["sample"] = line["sample"] + "; `value`: " + alt["sample"]
        del line["at_cgroup_limit"]
    else:
        line["value_is"] = "reference_rule (cores // 2 workers)"
    return line


# ---------------------------------------------------------------------------
# PMC traffic of the dominant kernel: a committed profiles/round<NN>_<workload>_pmc.json is used only while
# the kernel sources it was collected on are byte-identical to the ones this run loads.
# ---------------------------------------------------------------------------
def source_hashes():
    out = {}
    for f in sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.hpp"))):
        out[os.path.basename(f)] = hashlib.sha256(open(f, "rb").read()).hexdigest()[:16]
    return out


KERNEL_SOURCES = {"k_ssim_gauss": ["k_quality.hip"], "k_ssim_ffmpeg": ["k_quality.hip"], "k_dct8": ["k_dct8.hip", "vqa_math.hpp"],
                  "k_bgr2gray_hist": ["k_gray_hist.hip"], "k_canny_nms": ["k_canny.hip"], "k_block_sad": ["k_sad.hip"]}


def pmc_traffic(workload, kernel, frames_per_launch, default_mode):
    """(bytes per launch or None, traffic_source string)"""
    if not default_mode:
        return None, "none: PMC passes are collected for the default modes only"
    cands = []
    for p in glob.glob(os.path.join(REPO, "profiles", "round*_%s_pmc.json" % workload)):
        m = re.match(r"round(\d+)_", os.path.basename(p))
        if m:
            cands.append((int(m.group(1)), p))
    if not cands:
        return None, "none: no profiles/round*_%s_pmc.json" % workload
    _, path = max(cands)
    pmc = json.load(open(path))
    rel = os.path.relpath(path, REPO)
    want = pmc.get("source_sha256")
    if not want:
        return None, "stale: %s carries no source hashes (collected before the kernels were stamped)" % rel
    have = source_hashes()
    # the kernel's own file, the shared device header AND the orchestration (launch geometry, stream wiring, launcher
    # signatures): a PMC file collected before any of them changed is stale
    for f in KERNEL_SOURCES.get(kernel, []) + ["vqa_dev.hpp", "vqa_capi.hip", "vqa_kernels.hpp"]:
        if want.get(f) != have.get(f):
            return None, "stale: %s was collected on a different %s (@ %s)" % (rel, f, pmc.get("git_sha", "?"))
    for kname, ent in pmc["kernels"].items():
        if kname.split("<")[0].startswith(kernel):  # k_ssim_gauss -> k_ssim_gauss_p2<256>, k_dct8 -> k_dct8_march<..>, ...
            return int(ent["hbm_bytes"] * frames_per_launch / pmc["frames_per_launch"]), "%s @ %s" % (rel, pmc.get("git_sha", "?"))
    return None, "none: %s has no entry for %s" % (rel, kernel)


# ---------------------------------------------------------------------------
# torch.distributed bring-up for N > 1.  RCCL must work: a failure is fatal on every rank unless the ranks were
# deliberately pinned to one device (VQA_BENCH_DEVICE: the 1-GPU rehearsal), where RCCL refuses duplicate GPUs.
# ---------------------------------------------------------------------------
def init_dist(backend, rank, world, device, rehearsal, stub):
    """returns (td, backend_used, reduce_device, rccl_ranks)"""
    import datetime
    import torch
    import torch.distributed as td
    tmo = datetime.timedelta(seconds=300)
    if backend == "gloo" or rehearsal:
        # rehearsal = every rank pinned to ONE device (VQA_BENCH_DEVICE): RCCL refuses duplicate GPUs by design, so the
        # scalar reductions of the rehearsal go over gloo from the start (no half-built NCCL group to tear down)
        if rehearsal and backend != "gloo":
            sys.stderr.write("[bench] rank %d: rehearsal (VQA_BENCH_DEVICE=%d): ranks share one device, scalar reductions over gloo\n"
                             % (rank, device))
        td.init_process_group("gloo", rank=rank, world_size=world, timeout=tmo)
        return td, "gloo", "cpu", None
    try:  # device_id makes the RCCL communicator come up here, so a broken fabric shows now, on every rank
        td.init_process_group("nccl", rank=rank, world_size=world, timeout=tmo, device_id=torch.device("cuda", device))
        probe = torch.ones(1, device="cuda")
        td.all_reduce(probe)
        torch.cuda.synchronize()
        got = int(probe.item())
        if got != world:
            raise RuntimeError("probe all-reduce returned %d, expected %d" % (got, world))
        return td, "nccl", "cuda", got
    except Exception as e:
        sys.stderr.write("[bench] rank %d: FATAL: RCCL bring-up failed on device %d: %s\n" % (rank, device, e))
        sys.stderr.flush()
        os._exit(3)  # every rank takes this path (the failure is collective); no JSON line is printed


class StubEngine:
    """--stub-engine: stands in for the HIP engine so the N > 1 rank logic (barriers, max-over-ranks timing, scalar
    all-reduce, JSON shape) can be rehearsed on a CPU under gloo.  It computes nothing; the line says "stub": true."""

    def __init__(self, batch):
        self.batch = batch

    def step(self):
        time.sleep(0.002)
        return ({"ssim": np.full((self.batch, 3), 0.5)}, {"dct_energy": np.full(self.batch, 1.0)})


# ---------------------------------------------------------------------------
def end_to_end(rtvqa_amd, N, device, eng, ref_pin, dist_pin, h, w, full, mask, params, planes, smode, steps,
               yref_pin=None, ydist_pin=None):
    """The same workload fed from PINNED HOST memory every step: each buffer crosses PCIe once.  All uploads go through ONE
    context, the copy lane (vqa_copy_h2d on its stream: batches cross the link one after another), two more contexts
    measure alternate batches and wait for their upload on the device (vqa_stream_wait), three buffer sets: the upload of
    batch i+1 runs under the kernels of batch i - the pipeline of stream.py at the C ABI's level.
    bgr24: the reference and the distorted BGR streams cross.  yuv420p (yref_pin given): the distorted BGR stream (the
    complexity kernels' input) and BOTH planar streams (the quality kernels' inputs) cross; the reference BGR stream is
    not needed on the device at all."""
    from rtvqa_amd.engine import DeviceBuffer, DeviceFrames
    Be = dist_pin.shape[0] - 1
    fb = h * w * 3
    yuv = yref_pin is not None
    yb = yref_pin.shape[1] if yuv else 0
    engs = [eng, rtvqa_amd.Engine(device)]
    cp = rtvqa_amd.Engine(device)
    NB = 3
    dbufs = [(DeviceBuffer(eng, (yb if yuv else fb) * (Be + 1)), DeviceBuffer(eng, fb * (Be + 1)),
              DeviceBuffer(eng, yb * (Be + 1)) if yuv else None) for _ in range(NB)]

    def h2d(dst, src):
        N.check(cp.lib.vqa_copy_h2d(cp.ctx, dst.ptr, src.ctypes.data, src.nbytes), "h2d", cp.ctx)

    def upload(i):
        dr, dd, dyd = dbufs[i % NB]
        h2d(dd, dist_pin)
        if yuv:
            h2d(dr, yref_pin)
            h2d(dyd, ydist_pin)
        else:
            h2d(dr, ref_pin)

    def launch(i):
        e = engs[i & 1]
        dr, dd, dyd = dbufs[i % NB]
        e.wait_for(cp)
        fd = DeviceFrames(dd.ptr, Be + 1, h, w, owner=dd)
        if yuv:
            qr = DeviceFrames(dr.ptr + yb, Be, h, w, frame_stride=yb, row_stride=w, owner=dr, channels=1)
            qd = DeviceFrames(dyd.ptr + yb, Be, h, w, frame_stride=yb, row_stride=w, owner=dyd, channels=1)
        else:
            qr, qd = DeviceFrames(dr.ptr, Be + 1, h, w, owner=dr).slice(1, Be + 1), fd.slice(1, Be + 1)
        e.quality_submit(qr, qd, planes, smode)
        e.complexity_submit(fd.slice(1, Be + 1), fd.frame(0), mask, params)

    def wait(i):
        e = engs[i & 1]
        return e.quality_wait(), e.complexity_wait()

    def run(k):
        upload(0)
        for i in range(k):
            if i >= 2:
                wait(i - 2)      # frees its context and the buffer set batch i + 1 is about to be uploaded into
            launch(i)
            if i + 1 < k:
                upload(i + 1)
        for i in range(max(k - 2, 0), k):
            wait(i)

    run(3)  # warm both contexts (allocations, first-touch)
    t0 = time.perf_counter()
    run(steps)
    dt = time.perf_counter() - t0
    engs[1].close()
    cp.close()
    per_frame = (fb + 2 * yb) if yuv else 2 * fb
    gb = per_frame * (Be + 1) * steps / 1e9
    return dict(fps=round(Be * steps / dt, 1), h2d_GBps=round(gb / dt, 2), pinned=True,
                overlap="copy lane + 2 measuring contexts + 3 buffer sets: H2D of batch i+1 runs under the kernels of batch i",
                frames_per_step=Be, steps=steps, bytes_per_frame=per_frame,
                crossing=("distorted BGR24 + reference and distorted yuv420p" if yuv else "reference + distorted BGR24"),
                note="PCIe Gen5 x16 bound (%.1f MB per %dx%d frame); measured after the timed region, never `value`"
                     % (per_frame / 1e6, w, h))


# ---------------------------------------------------------------------------
def reduce_over_ranks(td, dt, pooled_vals, device, world, rehearsal, stub, red_dev):
    """max of the timed region over ranks, the path's ONE data collective (a SUM all-reduce of a few pooled float64
    scalars: RCCL over xGMI, latency-bound) and the device census.  td None = no process group.  -> (dt, devices)"""
    if td is None:
        return dt, [device]
    import torch
    tmax = torch.tensor([dt], dtype=torch.float64, device=red_dev)
    td.all_reduce(tmax, op=td.ReduceOp.MAX)
    dt = float(tmax.item())
    pooled = torch.tensor(pooled_vals, dtype=torch.float64, device=red_dev)
    td.all_reduce(pooled, op=td.ReduceOp.SUM)
    seen = [None] * world
    td.all_gather_object(seen, device)
    devices = sorted(set(int(x) for x in seen))
    if not rehearsal and not stub and len(devices) != world:
        sys.stderr.write("[bench] FATAL: %d ranks share %d devices %s\n" % (world, len(devices), devices))
        os._exit(3)
    return dt, devices


# ---------------------------------------------------------------------------
def kernel_report(prof, alg_bytes, valu_fma, mfma_flops, workload, frames_per_launch, default_mode):
    """prof: {kernel: (total ms, launches)} of a serial pass -> (per-kernel table, roofline of the dominant kernel).
    alg_bytes / valu_fma / mfma_flops: algorithmic bytes / counted fp32 FMAs / dense flops PER LAUNCH of the kernels that have such a
    figure (SURVEY.md 8d x the frames one launch covers)."""
    kernels = {}
    for name, (ms, cnt) in prof.items():
        per = ms / cnt
        ent = {"ms_per_launch": round(per, 4), "launches": cnt, "share_of_kernel_time": 0.0}
        if name in alg_bytes:
            gbs = alg_bytes[name] / (per * 1e-3) / 1e9
            ent.update({"alg_bytes": alg_bytes[name], "GBps": round(gbs, 1), "frac_hbm": round(gbs / HBM_PEAK_GBS, 4)})
        kernels[name] = ent
    tot = sum(ms for ms, _ in prof.values()) or 1.0
    for name, (ms, _) in prof.items():
        kernels[name]["share_of_kernel_time"] = round(ms / tot, 4)
    for name, fma in valu_fma.items():
        if name in kernels:
            tf = 2.0 * fma / (kernels[name]["ms_per_launch"] * 1e-3) / 1e12
            kernels[name].update({"fma": fma, "TFLOPps": round(tf, 1), "frac_fp32": round(tf / 157.3, 4)})
    for name, fl in mfma_flops.items():
        if name in kernels:
            tf = fl / (kernels[name]["ms_per_launch"] * 1e-3) / 1e12
            kernels[name].update({"flops": fl, "TFLOPps": round(tf, 1), "frac_mfma_f32": round(tf / 157.3, 4)})
    cands = [k for k in prof if k in alg_bytes or k in mfma_flops]
    dom = max(cands, key=lambda k: prof[k][0]) if cands else None
    if dom is None:
        return kernels, None
    if dom in mfma_flops:
        return kernels, {"bound": "mfma", "kernel": dom, "achieved": kernels[dom]["TFLOPps"], "peak": 157.3, "unit": "TFLOP/s",
                         "frac": kernels[dom]["frac_mfma_f32"], "traffic": None, "traffic_source": "none",
                         "note": "fp32-input MFMA (v_mfma_f32_32x32x2_f32); achieved = flops per launch / mean HIP-event duration"}
    traffic, tsrc = pmc_traffic(workload, dom, frames_per_launch, default_mode)
    if kernels[dom]["frac_hbm"] < 0.2 and "frac_fp32" in kernels[dom]:
        # a kernel this far below the HBM roof that does counted fp32 work is reported against the roof it
        # really has: the vector ALUs; the HBM figures stay beside it
        roof = {"bound": "valu_fp32", "kernel": dom, "achieved": kernels[dom]["TFLOPps"], "peak": 157.3,
                "unit": "TFLOP/s", "frac": kernels[dom]["frac_fp32"], "fma": kernels[dom]["fma"],
                "frac_hbm": kernels[dom]["frac_hbm"], "achieved_hbm_GBps": kernels[dom]["GBps"],
                "peak_hbm_GBps": HBM_PEAK_GBS, "traffic": traffic, "traffic_source": tsrc,
                "alg_bytes": alg_bytes[dom],
                "note": "achieved = 2 x 88 FMA per pixel and plane / mean HIP-event duration; fp32 vector peak "
                        "157.3 TFLOP/s at 2.4 GHz; the kernel issues packed FMAs at the calibrated rate "
                        "(DESIGN.md 4b/5) and the chip holds ~1.9-2.05 GHz under it (profiles/round*_clock.json); "
                        "frac_hbm = algorithmic bytes / time / 8 TB/s is kept for the HBM view"}
    else:
        roof = {"bound": "hbm", "kernel": dom, "achieved": kernels[dom]["GBps"], "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": kernels[dom]["frac_hbm"], "traffic": traffic, "traffic_source": tsrc,
                "alg_bytes": alg_bytes[dom],
                "note": "achieved = algorithmic bytes per launch / mean HIP-event duration of that launch"}
    return kernels, roof


# ---------------------------------------------------------------------------
API_LABELS = ("Advanced Motion Complexity", "DCT Complexity", "Temporal DCT Complexity", "Histogram Complexity",
              "Edge Detection Complexity", "ORB Feature Complexity", "Color Histogram Complexity", "Framerate Variation")
# (the reference unpacks the 8-tuple under these names in THIS order, video_processing.py:235-242 - which shifts five labels
#  against the tuple's real order, complexity_metrics.py:301-310; reading the row back in label order recovers the tuple)


def api_rates(vp, cm, clips, config, steps, frames):
    """process_video_and_extract_metrics (the reference's caller of run_ffmpeg_metrics + calculate_average_scene_complexity,
    video_processing.py:216, :242) on the same clip held three ways; one warm call (allocations, the pinned ring), then
    `steps` timed calls each.  clips: name -> (input, encoded) BGR pair, or (input, encoded, encoded_bgr) = a planar quality
    pair with the encoded stream's BGR frames beside it.  -> {name_fps: ...}, {name: the last call's row}"""
    import tempfile
    out, rows = {}, {}
    with tempfile.TemporaryDirectory() as tmp:
        csv = os.path.join(tmp, "bench_api.csv")
        for name, clip in clips.items():
            r, d = clip[0], clip[1]
            kw = {"encoded_bgr": clip[2]} if len(clip) > 2 else {}
            metrics = vp.process_video_and_extract_metrics(r, d, config, csv_file=csv, **kw)
            t0 = time.perf_counter()
            for _ in range(steps):
                metrics = vp.process_video_and_extract_metrics(r, d, config, csv_file=csv, **kw)
            out[name + "_fps"] = round(frames * steps / (time.perf_counter() - t0), 1)
            rows[name] = metrics
    return out, rows


def api_rows_check(cm, rows, crec, dist_clip, motion_mode=0):
    """The rows the entry point returned, against the C ABI's own (oracle-verified) records of the same frames: the last timed
    step measured frames 1..B of the clip against frames 0..B-1 - exactly the samples of the clip at frame_interval 1 - so
    pooling its records through the reference's tails must give the row's eight complexity values (to 1e-12: the entry point
    launches <= 100 frames at a time, the step 256 - and no record depends on that, Farneback's mean included since round 6),
    and the rows from every residence and caller thread must agree with each other to the last bit."""
    from rtvqa_amd import tails
    series = {k: tails.scalars(k, crec, motion_mode) for k in ("motion", "dct", "hist", "edge", "orb", "color")}
    series["temporal"] = tails.scalars("temporal", crec)[1:]
    want = cm.pool_series(series, dist_clip, 1)
    bad = []
    names = list(rows)
    for name in names:
        got = [rows[name][lab] for lab in API_LABELS]
        for k, (g, wv) in enumerate(zip(got, want)):
            g, wv = float(g), float(wv)
            if not ((g != g and wv != wv) or abs(g - wv) <= 1e-12 * max(abs(wv), 1e-300)):
                bad.append("%s tuple[%d] %.17g vs %.17g" % (name, k, g, wv))
        for lab in API_LABELS + ("PSNR", "SSIM"):
            a, b = rows[name][lab], rows[names[0]][lab]
            if a != b and not (a != a and b != b):
                bad.append("%s differs from %s in %s" % (name, names[0], lab))
    return bad


def api_end_to_end(clips, n, steps, value, e2e_fps, config, crec=None, dist_clip=None, motion_mode=0):
    """The drop-in Python surface on the measured path: resident / pinned host / pageable host clip of the workload's
    geometry through process_video_and_extract_metrics, in the workload's own modes (config keys ssim_mode, pixfmt, dct_mode,
    motion).  clips: name -> (input, encoded[, encoded_bgr]).  Measured after the timed region; never `value`."""
    from rtvqa_amd import complexity_metrics as cm
    from rtvqa_amd import video_processing as vp
    rates, rows = api_rates(vp, cm, clips, config, steps, n)
    # Two caller threads on the resident clip (the reference's surface is written for threaded callers, video_processing.py:25-41,
    # :56-67): their passes take turns on the device (stream.pass_lock), one's epilogue - tails of the last chunk, log parse, pooling,
    # CSV row - runs under the other's kernels
    import tempfile
    import threading
    with tempfile.TemporaryDirectory() as tmp:
        clip = clips["resident"]
        kw = {"encoded_bgr": clip[2]} if len(clip) > 2 else {}
        rows2, errs = [None, None], []

        def caller(t):
            try:
                for _ in range(steps):
                    rows2[t] = vp.process_video_and_extract_metrics(clip[0], clip[1], config, csv_file=os.path.join(tmp, "t.csv"), **kw)
            except BaseException as e:  # noqa: BLE001 - reported below
                errs.append(repr(e))
        ts = [threading.Thread(target=caller, args=(t,)) for t in range(2)]
        t0 = time.perf_counter()
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        dt2 = time.perf_counter() - t0
        with open(os.path.join(tmp, "t.csv")) as f:
            csv_rows = sum(1 for _ in f)
    tbad = []   # (reported through `verified` below: the caller exits after the ranks' final barrier, no rank is left hanging)
    if errs or csv_rows != 1 + 2 * steps or None in rows2:
        tbad.append("two caller threads: errors %s, %d CSV lines for %d rows" % (errs, csv_rows, 2 * steps))
        rows2 = [rows["resident"] if r is None else r for r in rows2]
    rates["resident_2_threads_fps"] = round(2 * steps * n / dt2, 1)
    rows["resident_thread_0"], rows["resident_thread_1"] = rows2
    cm.release_buffers()  # the pinned ring and lane buffers of the passes: given back (stream.release_buffers)
    m = rows["resident"]
    out = dict(rates, frames_per_call=n, calls=steps, config=config,
               entry_point="rtvqa_amd.video_processing.process_video_and_extract_metrics (= run_ffmpeg_metrics + "
                           "calculate_average_scene_complexity in ONE pass; stats files, regex parse and CSV row included)",
               PSNR=m.get("PSNR"), SSIM=m.get("SSIM"))
    if crec is not None or tbad:
        bad = tbad + (api_rows_check(cm, rows, crec, dist_clip, motion_mode) if crec is not None else [])
        out["verified"] = {"ok": not bad, "checker": "the C ABI's last timed step (itself verified against the oracle) pooled through the "
                                                     "reference's tails: the row's eight complexity values to 1e-12, the three residences and "
                                                     "two caller threads bit for bit"}
        if bad:  # (the caller prints no line and exits 4 after the ranks' final barrier: no rank is left hanging)
            sys.stderr.write("[bench] FATAL: the entry point's row differs from the C ABI's records: %s\n" % json.dumps(bad[:8]))
            sys.stderr.flush()
    if value:
        out["resident_vs_value"] = round(rates["resident_fps"] / value, 3)
        out["resident_2_threads_vs_value"] = round(rates["resident_2_threads_fps"] / value, 3)
    if e2e_fps:
        out["host_pinned_vs_end_to_end"] = round(rates["host_pinned_fps"] / e2e_fps, 3)
        out["host_pageable_vs_end_to_end"] = round(rates["host_pageable_fps"] / e2e_fps, 3)
    return out


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


# ---------------------------------------------------------------------------
# workload c1 = BASELINE.json configs[0]: the reference's own config.json on a short 1080p clip, through the
# reference-shaped entry points.
# ---------------------------------------------------------------------------
C1_CONFIG = {"crf": 23, "resize_width": 64, "resize_height": 64, "frame_interval": 10, "batch_size": 100}  # /root/reference/config.json
C1REF_KEYS = {"ssim_mode": "ffmpeg", "pixfmt": "yuv420p", "motion": "farneback"}  # (dct_mode auto = the reference's full-frame DCT at 64x64)


def _c1_quality_item(pair):
    from oracle import pipeline as pl
    from rtvqa_amd.engine import bgr_planes
    ref, dist = pair
    return pl.frame_quality(ref, dist, bgr_planes(ref.shape[0], ref.shape[1]), "gauss")


def _c1ref_quality_item(item):
    from oracle import pipeline as pl
    from rtvqa_amd.engine import yuv420p_planes
    ref, dist, h, w = item
    return pl.frame_quality(ref, dist, yuv420p_planes(h, w), "ffmpeg")


def c1_clip(rank, h, w, n):
    from rtvqa_amd import synth
    parts = [stream_chunk("natural", rank, h, w, a, min(CHUNK, n - a)) for a in range(0, n, CHUNK)]
    return np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts])


def c1_planar(clip):
    """the clip as the decoded yuv420p planes FFmpeg's filters would compare (frames.bgr_to_yuv420p, chunked)"""
    from rtvqa_amd.frames import bgr_to_yuv420p
    return np.concatenate([bgr_to_yuv420p(clip[a:a + CHUNK]) for a in range(0, len(clip), CHUNK)])


def c1_oracle(ref, dist, workers, pool=True, planar=None):
    """The reference's two calls on the clip, restated on the CPU: the quality filters over every frame pair (the ffmpeg
    subprocess of video_processing.py:270-297 - here the oracle's PSNR/SSIM under the same process pool) and
    calculate_average_scene_complexity with its own dispatcher (a NEW pool per metric pass, complexity_metrics.py:143).
    planar = (yref, ydist): the reference-true definitions (vf_ssim on the yuv420p planes, Farneback motion).
    -> (8-tuple, [(sse, ssim)] per frame)"""
    from concurrent.futures import ProcessPoolExecutor
    from oracle import pipeline as pl
    cfg = C1_CONFIG
    if planar is not None:
        h, w = dist.shape[1], dist.shape[2]
        items, fn = [(planar[0][i], planar[1][i], h, w) for i in range(len(ref))], _c1ref_quality_item
    else:
        items, fn = [(ref[i], dist[i]) for i in range(len(ref))], _c1_quality_item
    if pool:
        q = []
        with ProcessPoolExecutor(max_workers=workers) as ex:
            for i in range(0, len(items), 100):
                q.extend(ex.map(fn, items[i:i + 100]))
    else:
        q = [fn(it) for it in items]
    tup = pl.calculate_average_scene_complexity(list(dist), cfg["resize_width"], cfg["resize_height"], cfg["frame_interval"],
                                                num_workers=workers, batch_size=cfg["batch_size"], dct_mode="full",
                                                dispatcher=pl.process_in_batches if pool else pl.serial_map,
                                                motion="farneback" if planar is not None else "sad")
    return tup, q


def cpu_baseline_c1(ref, dist, planar=None):
    """ref/dist: the clip or its first frames (0.35 s of CPU work per frame pair for PSNR/SSIM, 0.36 s per selected frame
    for the complexity suite: the 300-frame clip is ~115 s of CPU work spread over the pool; reference-true definitions:
    7 ms per pair for psnr + vf_ssim on the planes, 0.7 s more per selected pair for Farneback: ~35 s)"""
    from oracle import c_oracle as co
    co.build()
    cores = visible_cores()
    workers, override = cpu_workers(cores)
    sample = len(ref)
    t0 = time.perf_counter()
    tup, q = c1_oracle(ref, dist, workers, planar=planar)
    dt = time.perf_counter() - t0
    # `cores` = the cores that actually ran it (finish_cpu_line); visible_cores = what os.sched_getaffinity shows the
    # process, which is what the reference's cpu_count() // 2 rule starts from
    line = dict(value=round(sample / dt, 3), unit="frames/s", cores=workers, visible_cores=cores, workers=workers, workers_override=override,
                cgroup_cpu_limit=cgroup_cpu_limit(), kind="port", seconds=round(dt, 2),
                sample="%d frames of the same clip through oracle/pipeline.py: %s of every frame "
                       "pair under ProcessPoolExecutor(max_workers=cores//2 = %d of %d visible cores), then "
                       "calculate_average_scene_complexity%s with the reference's dispatcher (a new pool per "
                       "metric pass, chunksize 1, batch_size 100)"
                       % (sample, "psnr + vf_ssim on the yuv420p planes" if planar is not None else "PSNR/SSIM", workers, cores,
                          " (Farneback motion)" if planar is not None else ""))

    def again(wk, m):
        t1 = time.perf_counter()
        c1_oracle(ref[:m], dist[:m], wk, planar=(planar[0][:m], planar[1][:m]) if planar is not None else None)
        return time.perf_counter() - t1
    # (the reference-true clip is ~35 s of CPU work in all: its second figure runs the whole sample, not half of it)
    line.update(second_cpu_figure(workers, override, again, min(sample, 200) if planar is None else 2 * sample))
    return (sample, tup, q), finish_cpu_line(line)


def main_c1(args, rank, local_rank, world):
    wl = WORKLOADS[args.workload]
    rt = bool(wl.get("ref_true"))  # c1ref: the reference's own definitions (planar quality pair + BGR stream, vf_ssim, Farneback)
    h, w = wl["h"], wl["w"]
    n = args.batch or wl["batch"]
    cfg = dict(C1_CONFIG, **(C1REF_KEYS if rt else {}))
    if args.stub_engine:
        raise SystemExit("--stub-engine rehearses the rank logic of the C-ABI workloads; %s has no stub" % args.workload)
    # ---- CPU first (its pools fork before this process touches the GPU): baseline on rank 0 at N = 1, and the expectation
    expect, cpu_line = None, None
    ref_h, dist_h = c1_clip(rank, h, w, n)  # ordinary host arrays: what a caller that decoded two files holds
    yref_h, ydist_h = (c1_planar(ref_h), c1_planar(dist_h)) if rt else (None, None)
    if world == 1 and rank == 0 and args.cpu_sample != 0:
        ns = min(n, args.cpu_sample if args.cpu_sample > 0 else n)  # default: the WHOLE clip (300 frames: ~105 s / ~35 s of CPU work over the pool)
        expect, cpu_line = cpu_baseline_c1(ref_h[:ns], dist_h[:ns], (yref_h[:ns], ydist_h[:ns]) if rt else None)
    elif args.verify:
        nv = min(n, 30)
        expect = (nv,) + c1_oracle(ref_h[:nv], dist_h[:nv], 1, pool=False, planar=(yref_h[:nv], ydist_h[:nv]) if rt else None)

    rehearsal = "VQA_BENCH_DEVICE" in os.environ
    device = int(os.environ.get("VQA_BENCH_DEVICE", local_rank))
    affinity = bind_numa(device, args.bind_numa)  # after the CPU baseline (it used every visible core), before any GPU call
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    torch.cuda.set_device(device)
    os.environ["VQA_DEVICE"] = str(device)  # the entry points' default engine (stream.get_engine)
    dist_on = world > 1 or args.dist_always
    td, backend_used, red_dev, rccl_ranks = None, None, "cpu", None
    if dist_on:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
        td, backend_used, red_dev, rccl_ranks = init_dist(args.backend, rank, world, device, rehearsal, False)

    import rtvqa_amd
    from rtvqa_amd import _native as N
    from rtvqa_amd import complexity_metrics as cm
    from rtvqa_amd import stream, synth
    from rtvqa_amd import video_processing as vp
    from rtvqa_amd.engine import DeviceFrames, bgr_planes, yuv420p_planes
    eng = cm.get_engine(device)
    if args.inflight == 1:  # the serial configuration (profiling runs): every chunk on the one default engine, in order
        stream.MAX_LANES = 1
    for e in stream.get_engine_pair(device):
        e.set_overlap(args.overlap)
    ref_pin, dist_pin = eng.alloc_pinned((n, h, w, 3)), eng.alloc_pinned((n, h, w, 3))
    ref_pin[...] = ref_h
    dist_pin[...] = dist_h
    dist_dev = eng.upload(dist_pin)
    if rt:
        yb = yref_h.shape[1]
        yref_pin, ydist_pin = eng.alloc_pinned((n, yb)), eng.alloc_pinned((n, yb))
        yref_pin[...] = yref_h
        ydist_pin[...] = ydist_h

        def planar_dev(a):
            d = eng.upload(a.reshape(n, 1, yb))
            return DeviceFrames(d.ptr, n, h, w, frame_stride=yb, row_stride=w, owner=d, channels=1)
        yref_dev, ydist_dev = planar_dev(yref_pin), planar_dev(ydist_pin)
        resident = (yref_dev, ydist_dev, dist_dev)
        host_clips = {"host_pinned": (yref_pin, ydist_pin, dist_pin), "host_pageable": (yref_h, ydist_h, dist_h)}
        planes, smode, motion_mode = yuv420p_planes(h, w), N.SSIM_FFMPEG, N.MOTION_FARNEBACK
    else:
        ref_dev = eng.upload(ref_pin)
        resident = (ref_dev, dist_dev)
        host_clips = {"host_pinned": (ref_pin, dist_pin), "host_pageable": (ref_h, dist_h)}
        planes, smode, motion_mode = bgr_planes(h, w), N.SSIM_GAUSS, N.MOTION_SAD
    import tempfile
    tmp = tempfile.mkdtemp(prefix="vqa_c1_")
    csv = os.path.join(tmp, "c1.csv")
    kw = {"encoded_bgr": resident[2]} if rt else {}

    def step():
        return vp.process_video_and_extract_metrics(resident[0], resident[1], cfg, csv_file=csv, **kw)

    def fence():
        for e in stream.get_engine_pair(device):
            e.sync()
        torch.cuda.synchronize()
        if dist_on:
            td.barrier()
        torch.cuda.synchronize()

    metrics = None
    for _ in range(max(args.warmup, 1)):
        metrics = step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        metrics = step()
    fence()
    dt = time.perf_counter() - t0

    # ---- serial pass for the per-kernel table: the same pass on ONE context, overlap off, HIP-event profiling on
    prof, dt_serial = {}, None
    if args.serial_pass and args.steps > 0:
        eng.set_overlap(False)
        qual = stream.Quality(planes, smode)
        cx = stream.Complexity((cfg["resize_width"], cfg["resize_height"]), cfg["frame_interval"], motion_mode=motion_mode)

        def serial():
            return stream.run(dist_dev, resident[0], qual, cx, cfg["batch_size"], engine=eng, qdist=resident[1] if rt else None)
        serial()
        eng.profile(True)
        eng.profile_read(reset=True)
        ts = time.perf_counter()
        for _ in range(args.steps):
            serial()
        eng.sync()
        dt_serial = time.perf_counter() - ts
        prof = eng.profile_read(reset=True)
        eng.profile(False)
        eng.set_overlap(args.overlap)

    # ---- verification: the entry points on the first nv frames against the oracle's numbers for those frames
    verified = None
    if args.verify and expect is not None:
        nv, tup, q = expect
        bad = []
        got = cm.calculate_average_scene_complexity(dist_dev.slice(0, nv), cfg["resize_width"], cfg["resize_height"],
                                                    frame_interval=cfg["frame_interval"], batch_size=cfg["batch_size"],
                                                    motion=cfg.get("motion"))
        for k, (g, wv) in enumerate(zip(got, tup)):
            g, wv = float(g), float(wv)
            tol = 1e-12 if k in (2, 3, 4, 5, 7) else 1e-4   # counts + the reference's own NumPy tails | DCT, motion floats
            if not ((g != g and wv != wv) or abs(g - wv) <= tol * max(abs(wv), 1e-30)):
                bad.append("tuple[%d] %.12g vs %.12g" % (k, g, wv))
        if rt:
            sse, ssim, sizes = vp.frame_quality(resident[0].slice(0, nv), resident[1].slice(0, nv), "yuv420p", "ffmpeg", h, w)
        else:
            sse, ssim, sizes = vp.frame_quality(resident[0].slice(0, nv), resident[1].slice(0, nv))
        for i, (es, em) in enumerate(q):
            if [int(v) for v in sse[i]] != [int(v) for v in es]:
                bad.append("frame %d sse %s != %s" % (i, sse[i].tolist(), es))
            if any(abs(float(a) - float(b)) > 1e-4 * abs(float(b)) for a, b in zip(ssim[i], em)):
                bad.append("frame %d ssim %s vs %s" % (i, ssim[i].tolist(), em))
        # the row the timed call returned: first-frame PSNR / SSIM as the stats files print them (area-weighted over the planes)
        es, em = q[0]
        areas = [float(sw * sh) for sw, sh in sizes]
        mse = sum(e / a * (a / sum(areas)) for e, a in zip(es, areas))
        allv = sum(v * (a / sum(areas)) for v, a in zip(em, areas))
        if abs(metrics["PSNR"] - 10.0 * np.log10(255.0 ** 2 / mse)) > 6e-3 or abs(metrics["SSIM"] - allv) > 1e-4:
            bad.append("row PSNR/SSIM %r %r" % (metrics["PSNR"], metrics["SSIM"]))
        verified = {"frames": nv, "ok": not bad, "checker": "oracle/pipeline.py on the first %d frames of the clip (computed before GPU init)" % nv,
                    "fields": "8-tuple: histogram/edge/orb/colour-histogram/frame-rate slots 1e-12 (exact counts + the reference's NumPy "
                              "tails), motion/dct/temporal 1e-4; per frame: sse exact, ssim 1e-4; the CSV row's PSNR/SSIM"}
        if bad:
            sys.stderr.write("[bench] rank %d: FATAL: %s output differs from the oracle: %s\n" % (rank, args.workload, json.dumps(bad[:8])))
            sys.stderr.flush()
        if dist_on:
            nbad = torch.tensor([float(len(bad))], dtype=torch.float64, device=red_dev)
            td.all_reduce(nbad, op=td.ReduceOp.SUM)
            if nbad.item() > 0:
                os._exit(4)
        elif bad:
            os._exit(4)

    dt, devices = reduce_over_ranks(td if dist_on else None, dt, [float(metrics.get("SSIM", 0.0)), float(metrics["DCT Complexity"]), float(n)],
                                    device, world, rehearsal, False, red_dev)
    value = n * args.steps * world / dt if args.steps > 0 else 0.0
    c1_bad = False
    if rank == 0:
        P = h * w
        # a chunk of the resident clip spans batch_size SAMPLES = batch_size * frame_interval source frames (stream.py)
        nchunks = max(1, -(-n // (cfg["batch_size"] * cfg["frame_interval"])))
        fpl = float(n) / nchunks  # frame pairs per quality launch
        pairs = max(len(cm.selected_indices(n, cfg["frame_interval"])) - 1, 0)
        if rt:
            # vf_ssim on yuv420p: three planar launches per chunk, 3P bytes in all -> P per frame and launch; Farneback: 279 B per
            # level-0 pixel and pair (the figure derived in main() for c3ref), ~pairs/chunks pairs per launch
            alg = {"k_ssim_ffmpeg": int(P * fpl), "farneback(pyramid)": int(279 * P * pairs / nchunks)}
            fma = {}
        else:
            alg = {"k_ssim_gauss": int(6 * P * fpl)}
            fma = {"k_ssim_gauss": int(88 * P * 3 * fpl)}
        kernels, roof = kernel_report(prof, alg, fma, {}, args.workload, fpl, False)
        line = {"metric": "frames/sec", "value": round(value, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
                "warmup": max(args.warmup, 1), "ms_per_step": round(dt / max(args.steps, 1) * 1e3, 4),
                "ms_per_step_serial": round(dt_serial / args.steps * 1e3, 4) if dt_serial else None,
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8->f32 (SSE/bins/SAD exact int)",
                "data": "synthetic (synth.s_natural v%d, seed 1234, one %d-frame clip per GPU; distorted = +-3 grey levels%s)"
                        % (synth.GENERATOR_VERSION, n, "; yuv420p planes derived by frames.bgr_to_yuv420p" if rt else ""),
                "config": {"workload": wl["name"], "id": args.workload, "frames_per_step_per_gpu": n, "reference_config": cfg,
                           "selected_frames": len(cm.selected_indices(n, cfg["frame_interval"])),
                           "entry_point": "rtvqa_amd.video_processing.process_video_and_extract_metrics"
                                          + ("(yref, ydist, config, encoded_bgr=...)" if rt else ""),
                           "collective": ("%s scalar all-reduce" % ("rccl" if backend_used == "nccl" else "gloo")) if dist_on else "none",
                           "backend": backend_used, "rccl_ranks": rccl_ranks, "devices": devices, "resident": "HBM",
                           "ssim_mode": cfg.get("ssim_mode", "gauss"), "pixfmt": cfg.get("pixfmt", "bgr24"),
                           "dct_mode": "auto (full-frame at 64x64, as cv2.dct)",
                           "motion": cfg.get("motion", "sad"), "parallelism": "1 clip/GPU x%d" % world,
                           "lanes": 1 if rt else stream.MAX_LANES,   # (Farneback: one context, stream.run)
                           "overlap": bool(args.overlap), "cpu_affinity": affinity,
                           "launched_by": "bench.py itself (--gpus N without a launcher)" if os.environ.get("VQA_BENCH_LAUNCHED") else
                                          ("torch.distributed.run" if "TORCHELASTIC_RUN_ID" in os.environ else "direct")},
                "roofline": roof, "kernels": kernels,
                "serial": {"ms_per_step": round(dt_serial / args.steps * 1e3, 4), "fps": round(n * args.steps / dt_serial, 1),
                           "what": "the same pass (stream.run) on ONE context, VQA_OPT_OVERLAP off, HIP-event profiling on"} if dt_serial else None,
                "row": {k: (float(v) if isinstance(v, (float, np.floating)) else v) for k, v in metrics.items()
                        if k in ("PSNR", "SSIM", "Resolution (px)")}}
        if args.api_steps > 0:
            rates, rows = api_rates(vp, cm, host_clips, cfg, args.api_steps, n)
            # (host chunks are capped in bytes where resident ones are not: no record follows the chunking, Farneback's included)
            same = all(rows[k][lab] == metrics[lab] for k in rows for lab in API_LABELS + ("PSNR", "SSIM"))
            if not same:
                sys.stderr.write("[bench] FATAL: %s rows from host memory differ from the resident clip's row\n" % args.workload)
                c1_bad = True
            per_frame = (2 * yref_h.shape[1] + 3 * P / cfg["frame_interval"]) if rt else 2 * 3 * P
            line["api_end_to_end"] = dict(rates, rows_identical_to_resident=same, resident_fps=round(value / world, 1), frames_per_call=n, calls=args.api_steps,
                                          bytes_per_frame=int(per_frame),
                                          note=("the same call from host memory: both yuv420p streams cross PCIe once (6.2 MB per frame pair) "
                                                "and every 10th BGR frame of the encoded stream (0.6 MB per source frame)") if rt else
                                               ("the same call from host memory: both 1080p streams cross PCIe once (12.4 MB per frame "
                                                "pair), the complexity kernels read every 10th frame of the uploaded chunk"))
        if cpu_line is not None:
            line["cpu_baseline"] = cpu_line
        if verified is not None:
            line["verified"] = verified
        if not c1_bad:  # no JSON line for a run whose output is wrong
            print(json.dumps(line), flush=True)
    if dist_on:
        td.barrier()
        td.destroy_process_group()
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
    cm.release_buffers()
    if c1_bad:
        os._exit(4)


# ---------------------------------------------------------------------------
# N > 1 without a launcher: `python bench.py --gpus N` starts the N ranks itself, as fresh children, and relays.
# ---------------------------------------------------------------------------
def launch_ranks(n):
    """No WORLD_SIZE in the environment and --gpus N > 1: this process has touched no GPU (nothing above imports torch.cuda or
    loads the library), so it starts `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same args>` as a
    CHILD process (never an exec: a restart of a process is not what a GPU box allows), lets rank 0's JSON line through on
    the inherited stdout and returns the launcher's exit code."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stderr.write("[bench] --gpus %d without a launcher: starting %d ranks: %s\n" % (n, n, " ".join(cmd)))
    sys.stderr.flush()
    env = dict(os.environ, VQA_BENCH_LAUNCHED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")   # (what torchrun would set with a warning)
    return subprocess.run(cmd, env=env).returncode


from rtvqa_amd.affinity import _parse_cpulist, bind_numa, gpu_local_cpus  # noqa: E402,F401  (sysfs only: no GPU call, no torch)


# ---------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS),
                    help="c3 (default) = BASELINE.json's headline: 1080p full suite + PSNR/SSIM; c2 = PSNR+SSIM+8x8 DCT only; "
                         "c4 = 2160p full suite; c3ref = the reference's own metric definitions (Farneback, full-frame DCT, vf_ssim on yuv420p)")
    ap.add_argument("--batch", type=int, default=0, help="frames per step per GPU (default: workload's)")
    ap.add_argument("--cpu-sample", type=int, default=-1, help="frame pairs for the CPU baseline (0 = skip, -1 = auto)")
    ap.add_argument("--e2e-steps", type=int, default=6, help="steps of the PCIe-inclusive end_to_end measurement (0 = skip)")
    ap.add_argument("--e2e-batch", type=int, default=64, help="frames per step of the end_to_end measurement")
    ap.add_argument("--api-steps", type=int, default=3,
                    help="timed calls per residence of the api_end_to_end measurement (the reference-shaped Python entry point on "
                         "the workload's clip; full-suite workloads in their default modes only; 0 = skip)")
    ap.add_argument("--dist-always", action="store_true",
                    help="initialise torch.distributed (and run every collective of the N > 1 path) even when WORLD_SIZE is 1: "
                         "the one-rank RCCL bring-up a 1-GPU box can execute")
    ap.add_argument("--ssim-mode", default=None, choices=["gauss", "ffmpeg"], help="default: gauss (c3ref: ffmpeg)")
    ap.add_argument("--streams", type=int, default=None, choices=[1, 2],
                    help="1 (default): quality and complexity kernels of a batch on one context; 2: the quality kernels on a "
                         "second context of the device (default for c3ref)")
    ap.add_argument("--inflight", type=int, default=None, choices=[1, 2, 3, 4],
                    help="batches in flight: step i+1 is submitted before step i is waited for, each on its own set of "
                         "contexts (default 2; 1 with Farneback motion - GiB-sized scratch per context)")
    ap.add_argument("--pixfmt", default=None, choices=["bgr24", "yuv420p"],
                    help="planes PSNR/SSIM compare: bgr24 (default: B,G,R of the packed frames) or yuv420p (Y + "
                         "quarter-size U,V derived from the same frames: what FFmpeg compares for an H.264 clip)")
    ap.add_argument("--content", default="natural", choices=["natural", "noise"],
                    help="synthetic stream: s_natural (default) or s_noise (max-entropy bins, worst case for Canny fan-out)")
    ap.add_argument("--dct-mode", default=None, choices=["block8", "full"],
                    help="block8 (default, north_star's 8x8 DCT) or full (the reference's full-frame cv2.dct, on fp32 MFMA)")
    ap.add_argument("--motion", default=None, choices=["sad", "farneback"],
                    help="motion metric of the full suite: sad (north_star's block-SAD, default) or farneback (the reference's own)")
    ap.add_argument("--no-overlap", dest="overlap", action="store_false",
                    help="turn VQA_OPT_OVERLAP off for the timed region (the library's default is on: block-SAD and the Canny "
                         "chain on side streams inside the complexity submit)")
    ap.add_argument("--no-serial-pass", dest="serial_pass", action="store_false",
                    help="skip the serial pass after the timed region (then the line carries no per-kernel table and no roofline)")
    ap.add_argument("--no-verify", dest="verify", action="store_false",
                    help="skip the post-timing check of the last timed step's records against the oracle")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N > 1 (nccl = RCCL over xGMI; gloo only to rehearse the rank logic)")
    ap.add_argument("--no-bind-numa", dest="bind_numa", action="store_false",
                    help="do not pin the rank to the CPUs of its GPU's NUMA node (default: pinned before any GPU call when "
                         "sysfs tells which CPUs those are; recorded as config.cpu_affinity)")
    ap.add_argument("--stub-engine", action="store_true",
                    help="CPU rehearsal of the N > 1 rank logic under gloo: no kernels run, the line carries \"stub\": true")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher: start the ranks as fresh children, relay, never touch a GPU here (an 8-GPU run that came back
        # with "n_gpus": 1 and rc 0 was the failure mode of round 5)
        raise SystemExit(launch_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.stub_engine and args.backend != "gloo":
        raise SystemExit("--stub-engine is a CPU rehearsal: use --backend gloo")

    wl = WORKLOADS[args.workload]
    for k, v in dict(dict(ssim_mode="gauss", pixfmt="bgr24", dct_mode="block8", motion="sad", streams=1), **wl.get("defaults", {})).items():
        if getattr(args, k) is None:
            setattr(args, k, v)
    if args.inflight is None:
        # Farneback keeps GiB-sized scratch per context and saturates the chip on its own: a second batch in flight
        # thrashed in round 3 (c3 + Farneback 2946 fps with one batch in flight, 1684 with two) and is within the
        # box-to-box noise with round 4's fused iteration (c3ref 4493 -> 4580 over 10 steps, 4435 -> 4354 over 3)
        args.inflight = 1 if args.motion == "farneback" else 2
    h, w, full = wl["h"], wl["w"], wl["full"]
    B = args.batch or wl["batch"]
    stub = args.stub_engine

    # CPU baseline first: its worker processes are forked before this process touches the GPU.  The same leg
    # computes the oracle's expected records for a few frames of this rank's stream (the checker of "verified").
    if wl.get("api"):   # c1 / c1ref: the reference's own config.json through the entry point
        return main_c1(args, rank, local_rank, world)
    cpu_line, expect = None, {}
    yuv = args.pixfmt == "yuv420p"
    if not stub and args.verify and args.steps > 0:
        vj = verify_frames(B) if rank == 0 else sorted(set([0, B - 1]))
        if args.motion == "farneback" or args.dct_mode == "full":  # the oracle needs seconds per 1080p frame here
            vj = sorted(set(j for j in (0, 1, B // 2, B - 1) if 0 <= j < B)) if rank == 0 else [0]
        vitems = [(args.content, rank, h, w, B, j, full, args.motion, args.ssim_mode, yuv, args.dct_mode) for j in vj]
        if world == 1 and args.cpu_sample != 0:
            expect, cpu_line = cpu_baseline(h, w, full, args.cpu_sample, args.motion, vitems)
        else:
            expect = dict(_expect_item(it) for it in vitems)
    elif world == 1 and rank == 0 and args.cpu_sample != 0 and not stub:
        _, cpu_line = cpu_baseline(h, w, full, args.cpu_sample, args.motion)

    rehearsal = "VQA_BENCH_DEVICE" in os.environ
    device = int(os.environ.get("VQA_BENCH_DEVICE", local_rank))
    # after the CPU baseline (it used every visible core), before any GPU call
    affinity = bind_numa(device, args.bind_numa) if not stub else {"bound": False, "why": "--stub-engine: no GPU to be near"}
    import torch
    if not stub:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
        # VQA_BENCH_DEVICE pins every rank to one device: only for rehearsing the N > 1 logic on a 1-GPU box
        torch.cuda.set_device(device)
    dist_on = world > 1 or args.dist_always
    td, backend_used, red_dev, rccl_ranks = None, None, "cpu", None
    if dist_on:
        if world == 1:  # --dist-always without a launcher: a rendezvous of one
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
        td, backend_used, red_dev, rccl_ranks = init_dist(args.backend, rank, world, device, rehearsal, stub)

    def sync_device():
        if not stub:
            torch.cuda.synchronize()

    if stub:
        eng = None
        all_engs = []
        se = StubEngine(B)
        step = se.step

        def fence():
            if dist_on:
                td.barrier()
        prof = {}
    else:
        import rtvqa_amd
        from rtvqa_amd import _native as N
        from rtvqa_amd import synth
        from rtvqa_amd.engine import DeviceBuffer, DeviceFrames, bgr_planes, yuv420p_planes
        from rtvqa_amd.frames import bgr_to_yuv420p, frame_bytes_yuv420p

        # contexts: a set = (complexity ctx, quality ctx); --streams 2 gives the quality kernels their own context (= their
        # own HIP stream on the same device), --inflight 2 a second set so step i+1 is enqueued while step i drains
        eng = rtvqa_amd.Engine(device)
        sets = []
        for k in range(args.inflight):
            ec = eng if k == 0 else rtvqa_amd.Engine(device)
            sets.append((ec, rtvqa_amd.Engine(device) if args.streams == 2 else ec))
        all_engs = []
        for ec, eq in sets:
            for e in (ec, eq):
                if e not in all_engs:
                    all_engs.append(e)
        for e in all_engs:
            e.set_overlap(args.overlap)

        # ---- synthetic streams, generated in chunks and made resident in HBM before timing
        fbytes = h * w * 3
        ref_buf, dist_buf = DeviceBuffer(eng, fbytes * (B + 1)), DeviceBuffer(eng, fbytes * (B + 1))
        ybytes = frame_bytes_yuv420p(h, w)
        if yuv:
            yref_buf, ydist_buf = DeviceBuffer(eng, ybytes * (B + 1)), DeviceBuffer(eng, ybytes * (B + 1))
        do_e2e = rank == 0 and args.e2e_steps > 0
        default_modes = (args.ssim_mode == "gauss" and not yuv and args.motion == "sad" and args.dct_mode == "block8")
        # (every full-suite workload in ITS OWN modes: the entry point takes them as config keys)
        do_api = rank == 0 and args.api_steps > 0 and full
        Be = min(args.e2e_batch, B)
        Bh = B if do_api else Be  # host copies: the whole clip for the API legs, else the end_to_end batch
        yref_pin = ydist_pin = None
        if do_e2e or do_api:  # page-locked host copies of the first Bh+1 frames (end_to_end reads the first Be+1 of them)
            ref_pin, dist_pin = eng.alloc_pinned((Bh + 1, h, w, 3)), eng.alloc_pinned((Bh + 1, h, w, 3))
            if yuv:  # the quality kernels' planar inputs cross PCIe too (the reference stream only in that form)
                yref_pin, ydist_pin = eng.alloc_pinned((Bh + 1, ybytes)), eng.alloc_pinned((Bh + 1, ybytes))
        for a in range(0, B + 1, CHUNK):
            n = min(CHUNK, B + 1 - a)
            r, d = stream_chunk(args.content, rank, h, w, a, n)
            N.check(eng.lib.vqa_copy_h2d(eng.ctx, ref_buf.ptr + a * fbytes, r.ctypes.data, r.nbytes), "h2d", eng.ctx)
            N.check(eng.lib.vqa_copy_h2d(eng.ctx, dist_buf.ptr + a * fbytes, d.ctypes.data, d.nbytes), "h2d", eng.ctx)
            if yuv:
                yr, yd = bgr_to_yuv420p(r), bgr_to_yuv420p(d)
                N.check(eng.lib.vqa_copy_h2d(eng.ctx, yref_buf.ptr + a * ybytes, yr.ctypes.data, yr.nbytes), "h2d", eng.ctx)
                N.check(eng.lib.vqa_copy_h2d(eng.ctx, ydist_buf.ptr + a * ybytes, yd.ctypes.data, yd.nbytes), "h2d", eng.ctx)
            eng.sync()
            if (do_e2e or do_api) and a <= Bh:
                m = min(n, Bh + 1 - a)
                ref_pin[a:a + m] = r[:m]
                dist_pin[a:a + m] = d[:m]
            if (do_e2e or do_api) and a <= Bh and yuv:
                m = min(n, Bh + 1 - a)
                yref_pin[a:a + m] = yr[:m].reshape(m, ybytes)
                ydist_pin[a:a + m] = yd[:m].reshape(m, ybytes)
        ref_all = DeviceFrames(ref_buf.ptr, B + 1, h, w, owner=ref_buf)
        dist_all = DeviceFrames(dist_buf.ptr, B + 1, h, w, owner=dist_buf)
        ref_b, dist_b, prev0 = ref_all.slice(1, B + 1), dist_all.slice(1, B + 1), dist_all.frame(0)

        mask = N.M_ALL if full else (N.M_DCT | N.M_TEMPORAL_DCT)
        params = eng.make_params(dct_mode=N.DCT_BLOCK8 if args.dct_mode == "block8" else N.DCT_FULL,
                                 motion_mode=N.MOTION_FARNEBACK if args.motion == "farneback" else N.MOTION_SAD)
        planes = bgr_planes(h, w)
        if yuv:  # quality kernels read the planar streams; the complexity kernels still read the BGR frames
            planes = yuv420p_planes(h, w)
            ref_b = DeviceFrames(yref_buf.ptr + ybytes, B, h, w, frame_stride=ybytes, row_stride=w, owner=yref_buf, channels=1)
            dist_q = DeviceFrames(ydist_buf.ptr + ybytes, B, h, w, frame_stride=ybytes, row_stride=w, owner=ydist_buf, channels=1)
        else:
            dist_q = dist_b
        smode = N.SSIM_GAUSS if args.ssim_mode == "gauss" else N.SSIM_FFMPEG

        def submit(i):
            ec, eq = sets[i % len(sets)]
            eq.quality_submit(ref_b, dist_q, planes, smode)
            ec.complexity_submit(dist_b, prev0, mask, params)

        def wait(i):
            ec, eq = sets[i % len(sets)]
            q = eq.quality_wait()
            return q, ec.complexity_wait()

        def run_steps(k):
            """k steps, at most len(sets) batches in flight; -> the records of the last step"""
            out = None
            for i in range(k):
                submit(i)
                if i >= len(sets) - 1:
                    out = wait(i - (len(sets) - 1))
            for i in range(max(k - (len(sets) - 1), 0), k):
                out = wait(i)
            return out

        def fence():
            for e in all_engs:
                e.sync()
            torch.cuda.synchronize()
            if dist_on:
                td.barrier()
            torch.cuda.synchronize()

    if stub:
        def run_steps(k):
            out = None
            for _ in range(k):
                out = step()
            return out

    run_steps(args.warmup)
    fence()
    t0 = time.perf_counter()
    last = run_steps(args.steps)
    fence()
    dt = time.perf_counter() - t0
    q, c = last if last is not None else (None, None)

    # ---- serial pass, outside the timed region: ONE context, overlap off, HIP-event profiling on.  Its per-kernel times
    # are free of GPU sharing and add up to (at most) its own step time; they fill "kernels" / "roofline".
    dt_serial = None
    serial_mismatch = False
    if not stub and args.serial_pass and args.steps > 0:
        eng.set_overlap(False)

        def serial_step():
            eng.quality_submit(ref_b, dist_q, planes, smode)
            eng.complexity_submit(dist_b, prev0, mask, params)
            return eng.quality_wait(), eng.complexity_wait()

        for _ in range(max(1, min(args.warmup, 2))):
            serial_step()
        eng.profile(True)
        eng.profile_read(reset=True)
        eng.sync()
        ts = time.perf_counter()
        for _ in range(args.steps):
            qs, cs = serial_step()
        eng.sync()
        dt_serial = time.perf_counter() - ts
        prof = eng.profile_read(reset=True)
        eng.profile(False)
        eng.set_overlap(args.overlap)
        if args.verify and not (np.array_equal(qs["sse"], q["sse"]) and np.array_equal(cs["edge_count"], c["edge_count"])
                                and np.array_equal(cs["sad_sum"], c["sad_sum"]) and np.array_equal(qs["ssim"], q["ssim"])
                                and np.array_equal(cs["dct_energy"], c["dct_energy"])):
            sys.stderr.write("[bench] rank %d: FATAL: the serial pass and the timed configuration disagree\n" % rank)
            serial_mismatch = True   # (fatal below, through the verification's collective: no rank is left waiting)
    elif not stub:
        prof = {}

    # ---- outside the timed region: the LAST TIMED step's records against the oracle's expectations
    verified = None
    if not stub and args.verify and args.steps == 0:
        verified = {"frames": [], "ok": None, "skipped": "--steps 0: no timed step to verify"}
    elif not stub and args.verify:
        from oracle import check  # checker only: the expectations were computed before the GPU was touched
        bad, notes = {}, []
        if serial_mismatch:
            bad["serial"] = ["the serial pass's records differ from the timed configuration's"]
        for j in sorted(expect):
            m = check.compare(expect[j], c[j], q[j], args.ssim_mode, notes)
            if m:
                bad[j] = m
        verified = {"frames": sorted(expect), "ok": not bad, "step": "last timed",
                    "fields": "sse exact, ssim 1e-4, "
                              + ("dct_energy/temporal_dct_l1 1e-4 (+Parseval)" if args.dct_mode == "block8" else
                                 "full-frame dct_energy 1e-4 vs Parseval, temporal_dct_l1 1e-4 vs scipy.fft.dctn float64")
                              + ((", edge count/strong/weak, %s, gray/B/G/R bins, orb count exact"
                                  % ("sad sum + mv histogram" if args.motion == "sad" else "Farneback flow_mag_mean 1e-4")) if full else ""),
                    "checker": "oracle/check.py (expected records computed on the host before GPU init)"}
        if notes:
            verified["looser_bar"] = notes
        if bad:
            sys.stderr.write("[bench] rank %d: FATAL: timed output differs from the oracle: %s\n" % (rank, json.dumps(bad)))
            sys.stderr.flush()
        if dist_on:
            nbad = torch.tensor([float(len(bad))], dtype=torch.float64, device=red_dev)
            td.all_reduce(nbad, op=td.ReduceOp.SUM)
            if nbad.item() > 0:
                os._exit(4)
            verified["ranks"] = world
        elif bad:
            os._exit(4)  # no JSON line for a run whose output is wrong

    # ---- max over ranks, and the one scalar all-reduce the path has (pooled metrics)
    dt, devices = reduce_over_ranks(td if dist_on else None, dt, [float(q["ssim"].mean()) if q is not None else 0.0,
                                                                  float(c["dct_energy"].mean()) if c is not None else 0.0, float(B)],
                                    device, world, rehearsal, stub, red_dev)
    frames_total = B * args.steps * world
    value = frames_total / dt if args.steps > 0 else 0.0
    api_bad = False

    if rank == 0:
        line = {
            "metric": "frames/sec", "value": round(value, 2), "unit": "frames/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / max(args.steps, 1) * 1e3, 4),
            "ms_per_step_serial": round(dt_serial / args.steps * 1e3, 4) if dt_serial else None,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8->f32 (SSE/bins/SAD exact int)",
        }
        config = {"workload": wl["name"] if args.ssim_mode == "gauss" else
                  wl["name"].replace("SSIM(gauss 11x11)", "SSIM(FFmpeg vf_ssim 8x8 integer)"),
                  "id": args.workload, "frames_per_step_per_gpu": B, "streams": args.streams, "inflight": args.inflight,
                  "collective": ("%s scalar all-reduce" % ("rccl" if backend_used == "nccl" else "gloo")) if dist_on else "none",
                  "backend": backend_used, "rccl_ranks": rccl_ranks, "devices": devices, "rehearsal_single_device": bool(rehearsal and world > 1),
                  "resident": "HBM", "ssim_mode": args.ssim_mode, "pixfmt": args.pixfmt, "dct_mode": args.dct_mode,
                  "motion": args.motion, "parallelism": "1 stream/GPU x%d" % world,
                  "overlap": bool(args.overlap), "cpu_affinity": affinity,
                  "launched_by": "bench.py itself (--gpus N without a launcher)" if os.environ.get("VQA_BENCH_LAUNCHED") else
                                 ("torch.distributed.run" if "TORCHELASTIC_RUN_ID" in os.environ else "direct")}
        if stub:
            line.update({"data": "STUB: no kernels ran (rank-logic rehearsal, --stub-engine)", "stub": True, "config": config,
                         "roofline": None, "kernels": {}})
            print(json.dumps(line), flush=True)
        else:
            P = h * w
            alg_bytes = {  # algorithmic HBM bytes per profiled launch group (SURVEY.md §8d x frames per launch)
                # bgr24: ONE launch covers the B, G, R planes (2P each).  yuv420p: two launches per step (Y: 2P,
                # U+V: 2 * 2 * P/4), so the mean launch moves 1.5P per frame
                "k_ssim_gauss": 2 * P * B * 3 if not yuv else int(1.5 * P * B),
                "k_ssim_ffmpeg": 2 * P * B * 3 if not yuv else P * B,  # yuv420p: three planar launches, 3P in all
                # the marching kernel serves BOTH DCT metrics from one read of every gray plane (+ prev0; + one halo
                # plane per internal chunk, not counted): P per frame.  SURVEY 8d's unfused figure (energy P +
                # temporal 2P) is kept beside it as unfused_alg_bytes, never used for frac_hbm
                "k_dct8": P * (B + 1), "k_bgr2gray_hist": 4 * P * (B + 1),
                "k_canny_nms": P * B + P * B // 4,  # reads gray, writes two bit-planes (P/8 each)
                "k_block_sad": 2 * P * B,
                # Farneback, per level-0 pixel and pair, every kernel reading its inputs and writing its outputs once:
                # level image 4 B + expansion 24 B + upsampled flow 10 B + 3 fused iterations x 56 B (R0 20 + R1 20 +
                # flow in 8 + flow out 8) = 206 B, x 4/3 for the pyramid, + 4 B of u8 reads (one per level) = 279 B
                # (the magnitude is summed by the last iteration; rounds 2-3, products through HBM: 440 B)
                "farneback(pyramid)": 279 * P * B,
            }
            def _smooth(n):  # the library's rule (k_dct_fft.hip, dct_fft_factor): even, 128..4000, prime factors 2, 3, 5
                if n < 128 or n > 4000 or n % 2:
                    return False
                for q in (2, 3, 5):
                    while n % q == 0:
                        n //= q
                return n == 1
            mfma_flops = {}
            if _smooth(h) and _smooth(w):
                # FFT-based row / column passes: both u8 planes read (2P), two float planes written (8P) and read back (8P)
                alg_bytes["k_dct_full"] = 18 * P * B
            else:
                mfma_flops = {"k_dct_full": 2 * 2.0 * (w * w * h + h * h * w) * B}  # energy + temporal, 2 dense products each
            # fp32 vector work of the Gaussian SSIM: separable 11+11 taps on 4 moment maps = 88 FMA per pixel and plane
            # (DESIGN.md section 5); the vector-ALU peak is 157.3 TFLOP/s (MI355X_MICROARCH.md)
            valu_fma = {"k_ssim_gauss": 88 * P * B * 3 if not yuv else int(88 * 1.5 * P * B / 2)}
            default_mode = (args.ssim_mode == "gauss" and not yuv and args.content == "natural" and args.motion == "sad"
                            and args.dct_mode == "block8")  # (the PMC passes profile the serial pass's launches)
            kernels, roof = kernel_report(prof, alg_bytes, valu_fma, mfma_flops, args.workload, B, default_mode)
            if "k_dct8" in kernels:
                kernels["k_dct8"]["unfused_alg_bytes"] = 3 * P * B
            serial = None
            if dt_serial:
                sum_ms = sum(ms for ms, _ in prof.values()) / args.steps
                serial = {"ms_per_step": round(dt_serial / args.steps * 1e3, 4), "sum_kernel_ms_per_step": round(sum_ms, 4),
                          "fps": round(B * args.steps / dt_serial, 1),
                          "what": "the same steps on ONE context, VQA_OPT_OVERLAP off, HIP-event profiling on, after the "
                                  "timed region: the source of `kernels` and `roofline` (event times free of GPU sharing)"}
            line.update({"data": "synthetic (synth.s_%s v%d, seed 1234, one stream per GPU; distorted = +-3 grey levels)"
                                 % (args.content, synth.GENERATOR_VERSION),
                         "config": config, "roofline": roof, "kernels": kernels, "serial": serial})
            if do_e2e:
                line["end_to_end"] = end_to_end(rtvqa_amd, N, device, eng, ref_pin[:Be + 1], dist_pin[:Be + 1], h, w, full, mask,
                                                params, planes, smode, args.e2e_steps,
                                                yref_pin[:Be + 1] if yuv else None, ydist_pin[:Be + 1] if yuv else None)
            if do_api:
                api_cfg = {"crf": 23, "resize_width": w, "resize_height": h, "frame_interval": 1, "batch_size": 100,
                           "ssim_mode": args.ssim_mode, "pixfmt": args.pixfmt, "dct_mode": args.dct_mode, "motion": args.motion}
                if yuv:  # the planar quality pair + the encoded stream's BGR frames, each held three ways
                    yref_all = DeviceFrames(yref_buf.ptr, B + 1, h, w, frame_stride=ybytes, row_stride=w, owner=yref_buf, channels=1)
                    ydist_all = DeviceFrames(ydist_buf.ptr, B + 1, h, w, frame_stride=ybytes, row_stride=w, owner=ydist_buf, channels=1)
                    clips = {"resident": (yref_all, ydist_all, dist_all), "host_pinned": (yref_pin, ydist_pin, dist_pin),
                             "host_pageable": (np.array(yref_pin), np.array(ydist_pin), np.array(dist_pin))}
                else:  # ordinary (pageable) copies, as a caller that decoded a file holds them
                    clips = {"resident": (ref_all, dist_all), "host_pinned": (ref_pin, dist_pin),
                             "host_pageable": (np.array(ref_pin), np.array(dist_pin))}
                line["api_end_to_end"] = api_end_to_end(clips, B + 1, args.api_steps, value, line.get("end_to_end", {}).get("fps"), api_cfg,
                                                        crec=c if (args.verify and args.steps > 0) else None, dist_clip=dist_all,
                                                        motion_mode=params.motion_mode)
            elif rank == 0 and args.api_steps > 0:
                line["api_end_to_end"] = None  # (c2 is not a full suite: process_video_and_extract_metrics always runs one)
            if cpu_line is not None:
                line["cpu_baseline"] = cpu_line
            if verified is not None:
                line["verified"] = verified
            api_bad = ((line.get("api_end_to_end") or {}).get("verified") or {}).get("ok") is False
            if not api_bad:  # no JSON line for a run whose output is wrong
                print(json.dumps(line), flush=True)
    if dist_on:
        td.barrier()
        td.destroy_process_group()
    for e in reversed(all_engs):
        e.close()
    if api_bad:
        os._exit(4)


if __name__ == "__main__":
    main()
