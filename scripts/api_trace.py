"""The reference-shaped entry point under a profiler: process_video_and_extract_metrics on a 1080p clip resident in HBM, from
pinned host memory and from pageable host memory (the three legs of bench.py's api_end_to_end), a few calls each.

  rocprofv3 --kernel-trace --stats -- python3 scripts/api_trace.py
      WHICH kernels the drop-in surface launches and how long they take there (profiles/round*_api_kernel_stats.csv)
  rocprofv3 --kernel-trace --memory-copy-trace --marker-trace --stats --output-format csv -d DIR -- python3 scripts/api_trace.py N pinned
      ONE residence, with the pass's roctx ranges (stream.py: vqa:gather / gather-wait / upload / submit / wait / tails per chunk
      and lane; "vqa:call" around each timed call): scripts/trace_summary.py turns DIR into H2D busy time, kernel busy time,
      their overlap and the idle gaps (profiles/round*_api_trace_<residence>.json)

usage: api_trace.py [frames=129] [resident|pinned|pageable|all] [calls=3]"""
import os, sys, tempfile, time
os.environ.setdefault("VQA_ROCTX", "1")   # markers on (a no-op without a marker library / profiler)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
if os.environ.get("VQA_BIND_NUMA", "1") != "0":   # before any GPU call: this process, its copier threads and its pinned pages next to the GPU
    from rtvqa_amd.affinity import bind_numa
    print("cpu affinity:", bind_numa(int(os.environ.get("VQA_DEVICE", "0"))), flush=True)
from rtvqa_amd import _native as N
from rtvqa_amd import complexity_metrics as cm, synth, video_processing as vp

n, h, w = int(sys.argv[1]) if len(sys.argv) > 1 else 129, 1080, 1920
which = sys.argv[2] if len(sys.argv) > 2 else "all"
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 3
eng = cm.get_engine()
pin_r, pin_d = eng.alloc_pinned((n, h, w, 3)), eng.alloc_pinned((n, h, w, 3))
for a in range(0, n, 32):
    r = synth.s_natural(min(32, n - a), h, w, seed=1234, t0=a)
    pin_r[a:a + len(r)] = r
    pin_d[a:a + len(r)] = synth.distort(r, t0=a)
clips = {}
if which in ("all", "resident"):
    clips["resident"] = (eng.upload(pin_r), eng.upload(pin_d))
if which in ("all", "pinned"):
    clips["host_pinned"] = (pin_r, pin_d)
if which in ("all", "pageable"):
    clips["host_pageable"] = (np.array(pin_r), np.array(pin_d))
cfg = {"crf": 23, "resize_width": w, "resize_height": h, "frame_interval": 1, "batch_size": 100}
csv = os.path.join(tempfile.gettempdir(), "api_trace.csv")
print("roctx markers: %s" % ("on" if N.roctx_active() else "off"), flush=True)
for name, (r, d) in clips.items():
    with N.trace_range("vqa:warm %s", name):
        m = vp.process_video_and_extract_metrics(r, d, cfg, csv_file=csv)   # allocations, the pinned ring, first-touch
    t0 = time.perf_counter()
    for i in range(calls):
        with N.trace_range("vqa:call %s %d", name, i):
            m = vp.process_video_and_extract_metrics(r, d, cfg, csv_file=csv)
    dt = time.perf_counter() - t0
    print("%-14s %d calls x %d frames in %.1f ms = %.0f frames/s  PSNR %.2f SSIM %.6f" % (name, calls, n, dt * 1e3, calls * n / dt, m["PSNR"], m["SSIM"]), flush=True)
cm.release_buffers()
