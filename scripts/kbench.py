#!/usr/bin/env python3
"""A/B timing of kernel variants (env VQA_SSIM_VARIANT / VQA_DCT_VARIANT), one child process per
variant so each gets a fresh library state.  usage: python scripts/kbench.py ssim 0 1 2 3 4 | dct 0 1 2 4 5 | dcte 0 1 2 4 5"""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def child(kind):
    import numpy as np
    import rtvqa_amd
    from rtvqa_amd import _native as N, synth
    from rtvqa_amd.engine import bgr_planes, gray_planes
    eng = rtvqa_amd.Engine(0)
    # (parity of every variant is covered by tests/ -m gpu, which reads the same env knobs)
    h, w, B = 1080, 1920, 64
    ref = synth.s_natural(B + 1, h, w, seed=1)
    dist = synth.distort(ref)
    dr, dd = eng.upload(ref), eng.upload(dist)
    eng.profile(True)
    for rep in range(6):
        if rep == 1:
            eng.profile_read(reset=True)
        if kind == "ssim":
            eng.quality(dr.slice(1, B + 1), dd.slice(1, B + 1), bgr_planes(h, w), N.SSIM_GAUSS)
        elif kind == "dcte":  # energy only (the c2 workload's launch)
            eng.complexity(dd.slice(1, B + 1), prev0=dd.frame(0), mask=N.M_DCT, dct_mode=N.DCT_BLOCK8)
        else:
            eng.complexity(dd.slice(1, B + 1), prev0=dd.frame(0), mask=N.M_DCT | N.M_TEMPORAL_DCT, dct_mode=N.DCT_BLOCK8)
    prof = eng.profile_read()
    for k, (ms, cnt) in prof.items():
        if k.startswith("k_ssim") or k.startswith("k_dct8"):
            per = ms / cnt
            npl = 3 if k.startswith("k_ssim") else 1
            nb = 1 if kind == "dcte" else 2  # algorithmic bytes per pixel
            print("  %-16s %.4f ms/launch (%d frames x %d planes) = %.3f us/frame-plane  %.0f GB/s" %
                  (k, per, B, npl, per * 1e3 / B / npl, nb * h * w * B * npl / (per * 1e-3) / 1e9), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(sys.argv[2])
    else:
        kind = sys.argv[1]
        for v in sys.argv[2:]:
            env = dict(os.environ)
            env.setdefault("VQA_LIB_PATH", os.path.join(REPO, "real-time-video-quality-analysis_amd", "csrc", "lab", "libvqa_hip_lab.so"))  # selectors: lab build only
            env["VQA_SSIM_VARIANT" if kind == "ssim" else "VQA_DCT_VARIANT"] = v
            print("%s variant %s" % (kind, v), flush=True)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", kind], env=env, timeout=300)
            if r.returncode:
                print("  FAILED rc=%d" % r.returncode, flush=True)
