#!/usr/bin/env python3
"""Third-party pins: scikit-image 0.18.3 / SciPy 1.7.1 outputs on seeded planes -> tests/golden/skimage_pins.json.

TEST INFRASTRUCTURE, build container only.  The image's system interpreter has no image library, but
/opt/conda/bin/python3.9 imports scikit-image 0.18.3 (with its own NumPy 1.26 / SciPy 1.7.1).  That is a library
nobody in this repository wrote, so what it computes pins the oracle where the two define the same quantity:

  * Gaussian-windowed SSIM as north_star defines it in place of video_processing.py:276
    (structural_similarity(gaussian_weights=True, sigma=1.5, use_sample_covariance=False, data_range=255):
    11x11 window, K1 = 0.01, K2 = 0.03, mean over the valid region)           -> oracle ssim_gauss, 1e-6
  * MSE / PSNR (video_processing.py:275 as FFmpeg's psnr filter defines them)  -> oracle sse_plane, exact / 1e-12
  * Shannon entropy of a plane (complexity_metrics.py:413-414, :467-473)       -> the product's f32 tails, 1e-6 / 1e-5
  * FAST-9/16 detection without NMS (corner_fast(n=9, threshold=20) on the integer-valued float image; the corner
    test cv2.ORB_create's detector applies, complexity_metrics.py:386-387)     -> oracle fast9(nonmax=False), exact map
  * 2x2 local means rounded half up (transform.downscale_local_mean): the arithmetic of the exact-halving shortcut of
    cv2.resize (INTER_LINEAR -> INTER_AREA at scale 2), pinned exactly;
  * float64 bilinear resize with half-pixel centres and edge clamping (transform.resize(order=1, mode="edge",
    anti_aliasing=False)): a GEOMETRY pin for cv2.resize INTER_LINEAR (complexity_metrics.py:359,404) - OpenCV's 11-bit
    fixed point must stay within 1 grey level of it; it does not pin OpenCV's rounding
  * BGR -> gray as ITU-R 601 luma of the right channels: Pillow 8.4.0's Image.convert("L") (16-bit fixed point,
    (R*19595 + G*38470 + B*7471 + 0x8000) >> 16) and the float64 formula 0.299 R + 0.587 G + 0.114 B.  A SEMANTIC pin
    for cv2.cvtColor(BGR2GRAY) (complexity_metrics.py:327,358,405,493): channel order and weights - OpenCV's 15-bit
    fixed point must stay within 1 grey level of both; it does not pin OpenCV's rounding   -> oracle bgr2gray, |diff| <= 1
  * Sobel 3x3 on a replicated border, L1 magnitude (scipy.ndimage.sobel(mode="nearest") on int32): the gradient
    stage of cv2.Canny(L2gradient=False) (complexity_metrics.py:503)           -> oracle sobel_l1, exact per pixel
  * orthonormal DCT-II (scipy.fft.dctn of SciPy 1.7.1): full-frame energy / L1 and the 8x8-block forms
    (complexity_metrics.py:363-364, :574-579)                                  -> oracle dct_*, 1e-5

Two stages, because the two interpreters cannot share a process: this file run under the system python3 writes the
seeded planes (rtvqa_amd.synth, integer-only generators) to a temp dir, re-runs itself under the conda interpreter
(`--stage2 DIR`), which writes results.json there, and then merges generator arguments, SHA-256 of every plane and
the values into the fixture.  Only arguments, hashes and numbers are committed: tests regenerate the planes and check
the hashes.  Nothing here is imported by the product.

    python oracle/gen_pins_skimage.py
"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONDA_PY = "/opt/conda/bin/python3.9"
FIXTURE = os.path.join(REPO, "tests", "golden", "skimage_pins.json")

# (name, generator, h, w, seed, channel, pairing)   pairing: "dist" = ref vs synth.distort(ref), "next" = frame t vs t+1
PAIR_CASES = [
    ("natural_72x104_B_dist", "natural", 72, 104, 11, 0, "dist"),
    ("natural_72x104_G_next", "natural", 72, 104, 11, 1, "next"),
    ("natural_97x131_G_dist", "natural", 97, 131, 12, 1, "dist"),
    ("natural_97x131_R_next", "natural", 97, 131, 12, 2, "next"),
    ("natural_270x480_R_dist", "natural", 270, 480, 13, 2, "dist"),
    ("natural_270x480_B_next", "natural", 270, 480, 13, 0, "next"),
    ("noise_64x80_B_dist", "noise", 64, 80, 14, 0, "dist"),
    ("noise_64x80_G_next", "noise", 64, 80, 14, 1, "next"),
    ("natural_11x11_B_dist", "natural", 11, 11, 15, 0, "dist"),       # one SSIM sample: the minimum plane
    ("natural_11x40_G_dist", "natural", 11, 40, 16, 1, "dist"),       # one row of samples
    ("natural_33x12_R_dist", "natural", 33, 12, 17, 2, "dist"),       # two columns of samples
    ("natural_1080x1920_B_dist", "natural", 1080, 1920, 1234, 0, "dist"),   # the bench stream's first B plane pair
    ("degenerate_48x64_zeros_same", "degenerate:zeros:same", 48, 64, 0, 0, ""),
    ("degenerate_48x64_full_vs_zeros", "degenerate:full:zeros", 48, 64, 0, 0, ""),
    ("degenerate_48x64_checker8_dist", "degenerate:checker8:dist", 48, 64, 0, 0, ""),
    ("degenerate_48x64_vstep_dist", "degenerate:vstep:dist", 48, 64, 0, 0, ""),
    ("degenerate_48x64_ramp_plus1", "degenerate:ramp:plus1", 48, 64, 0, 0, ""),
]
# BGR frames for the colour-entropy / gray-entropy / FAST / resize / DCT pins: (name, generator, h, w, seed)
FRAME_CASES = [
    ("natural_97x131", "natural", 97, 131, 12),
    ("natural_270x480", "natural", 270, 480, 13),
    ("noise_64x80", "noise", 64, 80, 14),
    ("natural_64x64", "natural", 64, 64, 18),
]
RESIZE_TARGETS = {"natural_97x131": (40, 24), "natural_270x480": (64, 64), "noise_64x80": (23, 17), "natural_64x64": (32, 48)}


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def make_pair(case):
    """-> (a, b): two contiguous 2-D uint8 planes.  Importable by the tests (system interpreter)."""
    sys.path.insert(0, REPO)
    from rtvqa_amd import synth
    name, gen, h, w, seed, ch, pairing = case
    if gen.startswith("degenerate"):
        _, which, other = gen.split(":")
        fr = synth.s_degenerate(h, w)
        a = fr[which][..., 0]
        if other == "same":
            b = a.copy()
        elif other == "dist":
            b = synth.distort(fr[which][None])[0][..., 0]
        elif other == "plus1":
            b = np.clip(a.astype(np.int16) + 1, 0, 255).astype(np.uint8)
        else:
            b = fr[other][..., 0]
        return np.ascontiguousarray(a), np.ascontiguousarray(b)
    fr = (synth.s_noise if gen == "noise" else synth.s_natural)(2, h, w, seed=seed)
    other = synth.distort(fr)[0] if pairing == "dist" else fr[1]
    return np.ascontiguousarray(fr[0][..., ch]), np.ascontiguousarray(other[..., ch])


def make_frames(case):
    """-> two consecutive BGR frames (n=2)."""
    sys.path.insert(0, REPO)
    from rtvqa_amd import synth
    name, gen, h, w, seed = case
    return (synth.s_noise if gen == "noise" else synth.s_natural)(2, h, w, seed=seed)


def gray_for(case):
    """The gray planes the frame pins are taken on: the oracle's BGR2GRAY of the two frames.  (The gray conversion
    itself is NOT pinned by this fixture; the pins hold for the plane whose SHA-256 is recorded.)"""
    from oracle import c_oracle as co
    fr = make_frames(case)
    return co.bgr2gray(fr[0]), co.bgr2gray(fr[1])


def stage1(tmp):
    sys.path.insert(0, REPO)
    meta = {"pairs": [], "frames": []}
    for case in PAIR_CASES:
        a, b = make_pair(case)
        np.save(os.path.join(tmp, case[0] + ".a.npy"), a)
        np.save(os.path.join(tmp, case[0] + ".b.npy"), b)
        meta["pairs"].append(dict(name=case[0], generator=case[1], h=case[2], w=case[3], seed=case[4], channel=case[5],
                                  pairing=case[6], sha_a=sha(a), sha_b=sha(b)))
    for case in FRAME_CASES:
        fr = make_frames(case)
        g0, g1 = gray_for(case)
        np.save(os.path.join(tmp, case[0] + ".bgr.npy"), fr[0])
        np.save(os.path.join(tmp, case[0] + ".g0.npy"), g0)
        np.save(os.path.join(tmp, case[0] + ".g1.npy"), g1)
        dw, dh = RESIZE_TARGETS[case[0]]
        meta["frames"].append(dict(name=case[0], generator=case[1], h=case[2], w=case[3], seed=case[4],
                                   sha_bgr=sha(fr[0]), sha_gray0=sha(g0), sha_gray1=sha(g1), resize_to=[dw, dh]))
    json.dump(meta, open(os.path.join(tmp, "meta.json"), "w"))


def stage2(tmp):
    """Runs under /opt/conda/bin/python3.9: scikit-image / SciPy only, nothing of this repository is imported."""
    import warnings
    warnings.filterwarnings("ignore")
    import scipy
    import scipy.fft
    import scipy.ndimage
    import PIL
    from PIL import Image
    import skimage
    from skimage.feature import corner_fast
    from skimage.measure import shannon_entropy
    from skimage.metrics import mean_squared_error, peak_signal_noise_ratio, structural_similarity
    from skimage.transform import resize, downscale_local_mean

    meta = json.load(open(os.path.join(tmp, "meta.json")))
    out = {"versions": {"skimage": skimage.__version__, "scipy": scipy.__version__, "numpy": np.__version__, "pillow": PIL.__version__,
                        "python": sys.version.split()[0]}, "pairs": {}, "frames": {}}
    for m in meta["pairs"]:
        a = np.load(os.path.join(tmp, m["name"] + ".a.npy"))
        b = np.load(os.path.join(tmp, m["name"] + ".b.npy"))
        mse = float(mean_squared_error(a, b))
        out["pairs"][m["name"]] = dict(
            ssim=float(structural_similarity(a, b, gaussian_weights=True, sigma=1.5, use_sample_covariance=False,
                                             data_range=255)),
            mse=mse,
            psnr=(float(peak_signal_noise_ratio(a, b, data_range=255)) if mse > 0 else "inf"),
            entropy_a=float(shannon_entropy(a)), entropy_b=float(shannon_entropy(b)))

    def blocks8(x):
        h, w = x.shape
        p = np.zeros(((h + 7) // 8 * 8, (w + 7) // 8 * 8))
        p[:h, :w] = x
        return p.reshape(p.shape[0] // 8, 8, p.shape[1] // 8, 8).transpose(0, 2, 1, 3)

    for m in meta["frames"]:
        bgr = np.load(os.path.join(tmp, m["name"] + ".bgr.npy"))
        g0 = np.load(os.path.join(tmp, m["name"] + ".g0.npy")).astype(np.float64)
        g1 = np.load(os.path.join(tmp, m["name"] + ".g1.npy")).astype(np.float64)
        dw, dh = m["resize_to"]
        fast = corner_fast(g0, n=9, threshold=20) > 0
        ys, xs = np.nonzero(fast)
        d0, d1 = scipy.fft.dctn(g0, norm="ortho"), scipy.fft.dctn(g1, norm="ortho")
        b0 = scipy.fft.dctn(blocks8(g0), axes=(2, 3), norm="ortho")
        b1 = scipy.fft.dctn(blocks8(g1), axes=(2, 3), norm="ortho")
        rs = resize(g0, (dh, dw), order=1, mode="edge", anti_aliasing=False, preserve_range=True)
        rgb = np.ascontiguousarray(bgr[..., ::-1])
        lum = np.asarray(Image.fromarray(rgb, "RGB").convert("L")).astype(np.int64)
        lum_f = 0.299 * rgb[..., 0].astype(np.float64) + 0.587 * rgb[..., 1] + 0.114 * rgb[..., 2]
        gi = g0.astype(np.int32)
        sob = np.abs(scipy.ndimage.sobel(gi, axis=1, mode="nearest")) + np.abs(scipy.ndimage.sobel(gi, axis=0, mode="nearest"))
        sh, sw = sob.shape
        wts = (np.arange(sh, dtype=np.int64)[:, None] * 31 + np.arange(sw, dtype=np.int64)[None, :] * 17 + 1) % 1009
        # exact halving: cv2.resize(INTER_LINEAR) takes INTER_AREA's 2x2 mean there, (a + b + c + d + 2) >> 2 = floor(mean + 0.5)
        he, we = g0.shape[0] & ~1, g0.shape[1] & ~1
        a2 = np.floor(downscale_local_mean(g0[:he, :we], (2, 2)) + 0.5).astype(np.int64)
        w2 = (np.arange(he // 2, dtype=np.int64)[:, None] * 31 + np.arange(we // 2, dtype=np.int64)[None, :] * 17 + 1) % 1009
        out["frames"][m["name"]] = dict(
            gray_entropy=float(shannon_entropy(g0.astype(np.uint8))),
            color_entropy_sum=float(sum(shannon_entropy(bgr[..., c]) for c in range(3))),
            fast9_count=int(fast.sum()),
            fast9_crc=int(np.sum((ys.astype(np.int64) * 7919 + xs.astype(np.int64) * 104729) % 1000003)),
            dct_full_energy=float((d1 ** 2).sum()), dct_full_l1=float(np.abs(d0 - d1).sum()),
            dct8_energy=float((b1 ** 2).sum()), dct8_l1=float(np.abs(b0 - b1).sum()),
            pillow_luma_sum=int(lum.sum()), pillow_luma_crc=int((lum * ((np.arange(lum.size, dtype=np.int64).reshape(lum.shape) % 251) + 1)).sum()),
            float_luma_sum=float(lum_f.sum()),
            gray_vs_pillow_maxdiff=int(np.abs(lum - g0.astype(np.int64)).max()), gray_vs_pillow_ndiff=int((lum != g0.astype(np.int64)).sum()),
            gray_vs_float_maxdiff=float(np.abs(lum_f - g0).max()),
            sobel_l1_sum=int(sob.sum()), sobel_l1_max=int(sob.max()), sobel_l1_crc=int((sob.astype(np.int64) * wts).sum()),
            sobel_l1_border_sum=int(sob[0].sum() + sob[-1].sum() + sob[:, 0].sum() + sob[:, -1].sum()),
            area2_sum=int(a2.sum()), area2_max=int(a2.max()), area2_crc=int((a2 * w2).sum()),
            resize_float=[round(float(v), 6) for v in rs.ravel()])
    json.dump(out, open(os.path.join(tmp, "results.json"), "w"))


def main():
    if len(sys.argv) == 3 and sys.argv[1] == "--stage2":
        return stage2(sys.argv[2])
    with tempfile.TemporaryDirectory() as tmp:
        stage1(tmp)
        env = {k: v for k, v in os.environ.items() if k not in ("PYTHONPATH", "PYTHONHOME")}
        subprocess.check_call([CONDA_PY, os.path.abspath(__file__), "--stage2", tmp], env=env, cwd=tmp)
        meta = json.load(open(os.path.join(tmp, "meta.json")))
        res = json.load(open(os.path.join(tmp, "results.json")))
    for m in meta["pairs"]:
        m.update(res["pairs"][m["name"]])
    for m in meta["frames"]:
        m.update(res["frames"][m["name"]])
    from rtvqa_amd import synth
    json.dump({"note": "outputs of scikit-image / SciPy (a third-party library, versions below) on seeded planes; made by "
                       "oracle/gen_pins_skimage.py in the build container; planes are regenerated by the tests and "
                       "checked against the SHA-256 recorded here",
               "versions": res["versions"], "synth_generator_version": synth.GENERATOR_VERSION,
               "ssim_call": "skimage.metrics.structural_similarity(a, b, gaussian_weights=True, sigma=1.5, "
                            "use_sample_covariance=False, data_range=255)",
               "pairs": meta["pairs"], "frames": meta["frames"]}, open(FIXTURE, "w"), indent=1)
    print("wrote", FIXTURE, "(%d pairs, %d frames)" % (len(meta["pairs"]), len(meta["frames"])))


if __name__ == "__main__":
    main()
