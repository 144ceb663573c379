"""Generate tests/golden/host_math.json by running the REAL reference's cv2-free
host functions (run in the build container only; /root/reference does not
travel to the GPU box, the JSON does).

The reference's pixel path cannot be imported: `import cv2` raises
ModuleNotFoundError (opencv_python is not installed, no network).  An EMPTY
module object registered as sys.modules["cv2"] lets the reference's modules
import; only functions that never touch cv2 are then called, unchanged:

  complexity_metrics.smooth_data                        (:114-125)
  complexity_metrics.process_in_batches                 (:128-148)
  complexity_metrics.process_frame_interval_for_parallel(:150-165)
  complexity_metrics.normalize                          (:167-169)
  complexity_metrics.calculate_scene_complexity_score   (:171-242, with
      calculate_average_scene_complexity replaced by a constant tuple so only the
      min/max table and the weights run)
  video_processing.extract_metrics_from_logs            (:145-177)

Inputs and outputs are data; no reference source text is copied.
Usage: python oracle/gen_golden.py
"""
import json
import os
import sys
import tempfile
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(REPO, "tests", "golden", "host_math.json")


def main():
    if not os.path.isdir("/root/reference"):
        sys.exit("/root/reference not present: golden vectors can only be generated in the build container")
    work = tempfile.mkdtemp()
    os.chdir(work)  # the reference opens video_processing.log in the cwd at import
    sys.modules["cv2"] = types.ModuleType("cv2")
    sys.path.insert(0, "/root/reference")
    import complexity_metrics as ref  # noqa: E402
    import video_processing as refvp  # noqa: E402

    rng = np.random.default_rng(20241016)
    g = {"generator": "oracle/gen_golden.py", "reference": "zaki699/Real-Time-Video-Quality-Analysis @ 2024-10-16",
         "numpy": np.__version__}
    import pandas
    g["pandas"] = pandas.__version__

    # --- smooth_data + np.mean (a3)
    series = [
        [1, 4, 2, 8, 5],
        [0.0],
        [3.5, 3.5, 3.5, 3.5],
        list(map(float, rng.normal(1000, 300, 29))),
        list(map(float, rng.integers(0, 4096, 299))),
        [2257755.5, 2251000.25, 2300123.0, 1999999.0, 2257755.5, 2100000.0],
        [],
    ]
    cases = []
    for s in series:
        for alpha in (0.8, 0.5, 0.2, 1.0):
            sm = ref.smooth_data(s, alpha)
            with np.errstate(all="ignore"):
                import warnings
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    mean = float(np.mean(sm))
            cases.append({"data": s, "alpha": alpha, "smoothed": [float(v) for v in sm],
                          "mean": None if np.isnan(mean) else mean})
    g["smooth_data"] = cases

    # --- process_in_batches ordering (a2); abs is picklable
    pib = []
    for items, bs, workers in (([-3, -2, -1, 0, 1, 2, 3], 3, 2), (list(range(-10, 11)), 4, 3), ([], 5, 2),
                               ([-1.5, 2.5], 100, 1)):
        pib.append({"items": items, "batch_size": bs, "num_workers": workers,
                    "out": ref.process_in_batches(items, abs, workers, batch_size=bs)})
    g["process_in_batches_abs"] = pib

    # --- frame interval -> fps
    fi = []
    for pair in ((0.0, 333.3333), (5.0, 5.0), (100.0, 90.0), (0.0, 1000.0 / 30 * 10), (33.366666, 66.733333)):
        fi.append({"timestamps": list(pair), "out": ref.process_frame_interval_for_parallel(pair)})
    g["process_frame_interval"] = fi

    # --- normalize
    nm = []
    for v, lo, hi in ((2257755.44, 1e6, 5e7), (3.0, 0.0, 2.0), (1.0, 2.0, 2.0), (-1.0, 0.0, 10.0), (177.94, 0.0, 1.0)):
        nm.append({"args": [v, lo, hi], "out": ref.normalize(v, lo, hi)})
    g["normalize"] = nm

    # --- weighted score: min/max table + weights only
    sc = []
    tuples = [
        # README.md:72 sample row in complexity_metrics.py:301-310 order
        (1.8996, 2257755.44, 2.7019, 177.94, 0.0810, 8.0762, 235402.93, 3.0),
        (0.0, 1e6, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0),
        (10.0, 5e7, 8.0, 1.0, 5000.0, 8.0, 1e7, 2.0),
        tuple(float(v) for v in rng.uniform(0, 10, 8)),
    ]
    orig = ref.calculate_average_scene_complexity
    for t in tuples:
        ref.calculate_average_scene_complexity = lambda *a, _t=t, **k: _t
        sc.append({"metrics_tuple": list(t), "score": float(ref.calculate_scene_complexity_score("x.mp4", 64, 64))})
    ref.calculate_average_scene_complexity = orig
    g["scene_complexity_score"] = sc

    # --- extract_metrics_from_logs on stats text in FFmpeg's format
    logs = []
    samples = [
        ("n:1 mse_avg:0.52 mse_y:0.62 mse_u:0.33 mse_v:0.31 psnr_avg:50.98 psnr_y:50.21 psnr_u:52.98 psnr_v:53.21 \n"
         "n:2 mse_avg:0.60 mse_y:0.70 mse_u:0.40 mse_v:0.41 psnr_avg:50.35 psnr_y:49.68 psnr_u:52.11 psnr_v:52.00 \n",
         "n:1 Y:0.995123 U:0.993000 V:0.994100 All:0.994599 (22.675034)\nn:2 Y:0.99 U:0.99 V:0.99 All:0.990000 (20.000000)\n"),
        ("n:1 mse_avg:0.00 mse_r:0.00 mse_g:0.00 mse_b:0.00 psnr_avg:inf psnr_r:inf psnr_g:inf psnr_b:inf \n",
         "n:1 R:1.000000 G:1.000000 B:1.000000 All:1.000000 (inf)\n"),
        ("n:1 mse_avg:1.00 mse_r:1.00 mse_g:1.00 mse_b:1.00 psnr_avg:48.13 psnr_r:48.13 psnr_g:48.13 psnr_b:48.13 \n",
         "n:1 R:0.912345 G:0.900000 B:0.800000 All:0.870782 (8.886765)\n"),
    ]
    for k, (ptxt, stxt) in enumerate(samples):
        pl, sl = os.path.join(work, "p%d.log" % k), os.path.join(work, "s%d.log" % k)
        open(pl, "w").write(ptxt)
        open(sl, "w").write(stxt)
        m = refvp.extract_metrics_from_logs(pl, sl, os.path.join(work, "absent.json"), "in.mp4", 23, 4486,
                                            "1920x1080", 30.0)
        logs.append({"psnr_text": ptxt, "ssim_text": stxt, "metrics": m})
    g["extract_metrics_from_logs"] = logs
    try:
        refvp.listener.stop()
    except Exception:
        pass

    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, "w") as f:
        json.dump(g, f, indent=1)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
