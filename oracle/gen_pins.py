#!/usr/bin/env python3
"""Freeze the oracle's outputs on small seeded frames into tests/golden/oracle_pins.json.

These are REGRESSION PINS OF THE RESTATEMENT, not outputs of the reference (which cannot run here: no cv2,
no ffmpeg — DESIGN.md §3): they make an accidental edit of oracle/vqa_oracle.c visible.  Re-run only when the
restatement is changed on purpose:  python oracle/gen_pins.py

History of deliberate regenerations (the values are the oracle pinning ITSELF, so each one is recorded here):
  round 3  farneback: fb_blur_solve rewritten in FarnebackUpdateFlow_Blur's own summation order (~1e-8 relative);
           still a self-pin - cv2.calcOpticalFlowFarneback cannot run here.
  round 4  ssim_gauss_b: Gaussian taps kept in double instead of rounded to float32 (~1.5e-8 relative) after
           scikit-image disagreed by 3.1e-6 on full-white vs full-black (tests/golden/skimage_pins.json is the
           third-party pin for that function; this file is not).
"""
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from oracle import c_oracle as co  # noqa: E402
from rtvqa_amd import synth  # noqa: E402


def cases():
    out = []
    for name, (h, w), seed in (("natural_96x128", (96, 128), 3), ("natural_67x131", (67, 131), 4), ("noise_64x64", (64, 64), 5)):
        fr = synth.s_noise(3, h, w, seed=seed) if name.startswith("noise") else synth.s_natural(3, h, w, seed=seed)
        dist = synth.distort(fr)
        g = [co.bgr2gray(f) for f in dist]
        rs = co.resize_linear(dist[1], 40, 24)
        e, l1, _ = co.dct8x8(g[0], g[1])
        nb, sad, hist = co.block_sad(g[0], g[1], 7)
        out.append(dict(
            name=name, h=h, w=w, seed=seed,
            gray_sum=int(g[1].astype(np.int64).sum()), gray_sq=int((g[1].astype(np.int64) ** 2).sum()),
            resize_bgr_sum=int(rs.astype(np.int64).sum()),
            resize_gray_sum=int(co.resize_linear(g[1], 40, 24).astype(np.int64).sum()),
            hist_gray_crc=int(np.dot(co.hist_u8(g[1]).astype(np.int64), np.arange(1, 257))),
            canny_100_200=list(map(int, co.canny(g[1], 100, 200))), canny_20_60=list(map(int, co.canny(g[1], 20, 60))),
            dct8_energy=e, dct8_l1=l1, dct_full_energy=co.dct_energy_full(g[1]), dct_full_l1=co.temporal_dct_full(g[0], g[1]),
            sad=[int(nb), int(sad)], sad_hist_crc=int(np.dot(hist.astype(np.int64), np.arange(1, 130))),
            sse_b=int(co.sse_plane(fr[1][..., 0], dist[1][..., 0])),
            ssim_gauss_b=co.ssim_gauss(fr[1][..., 0], dist[1][..., 0]), ssim_ffmpeg_b=co.ssim_ffmpeg(fr[1][..., 0], dist[1][..., 0]),
            orb=list(map(int, co.orb64_count(co.bgr2gray(co.resize_linear(dist[1], 64, 64))))),
            fast9=int(co.fast9(g[1], 20, True)[0]),
            farneback=co.farneback(g[0], g[1]),
        ))
    return out


if __name__ == "__main__":
    path = os.path.join(REPO, "tests", "golden", "oracle_pins.json")
    json.dump({"note": "regression pins of the oracle restatement (NOT reference outputs); made by oracle/gen_pins.py",
               "cases": cases()}, open(path, "w"), indent=1)
    print("wrote", path)
