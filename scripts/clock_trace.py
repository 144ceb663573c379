"""Shader clock / power while each hot kernel runs alone and while the whole c3 step runs (VERDICT round 2 #5:
"the chip holds 1.70 GHz under this load" as evidence, not prose).

A sampler thread polls `rocm-smi --showclocks --showpower --json` (falls back to text parsing) while the main thread
keeps one workload queued on the GPU for a few seconds; per phase the script keeps every sample and reports
min / median / max of sclk and the socket power.  Output: gpurun_out/clock_trace.json (copy to profiles/).
usage (GPU box): python scripts/clock_trace.py [seconds_per_phase]
(measurement tool, not product code)
"""
import json
import os
import re
import subprocess
import sys
import threading
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402
import rtvqa_amd  # noqa: E402
from rtvqa_amd import _native as N, synth  # noqa: E402
from rtvqa_amd.engine import bgr_planes  # noqa: E402

SMI = "/opt/rocm/bin/rocm-smi"
samples, phase, stop = [], ["idle"], [False]


def poll():
    while not stop[0]:
        t = time.time()
        try:
            out = subprocess.run([SMI, "-d", "0", "--showclocks", "--showpower", "--json"], capture_output=True, text=True,
                                 timeout=10).stdout
            rec = {"t": t, "phase": phase[0]}
            try:
                card = next(iter(json.loads(out).values()))
                for k, v in card.items():
                    kl = k.lower()
                    m = re.search(r"\((\d+)Mhz\)", str(v))
                    if kl.startswith("sclk") and m:
                        rec["sclk_mhz"] = int(m.group(1))
                    elif kl.startswith("mclk") and m:
                        rec["mclk_mhz"] = int(m.group(1))
                    elif "power" in kl and "(w)" in kl:
                        try:
                            rec["power_w"] = float(v)
                        except ValueError:
                            pass
            except Exception:
                m = re.search(r"sclk[^\n]*\((\d+)Mhz\)", out)
                if m:
                    rec["sclk_mhz"] = int(m.group(1))
                rec["raw"] = out[:400]
            samples.append(rec)
        except Exception as e:  # keep sampling
            samples.append({"t": t, "phase": phase[0], "error": str(e)})
        time.sleep(0.05)


def power_cap():
    """What the box says its power limit is ("no power_cap_w read from the box is recorded anywhere"):
    rocm-smi's view and the hwmon files of every amdgpu card, raw.  An ordinary user can read these; none is changed."""
    out = {}
    for flag in ("--showmaxpower", "--showpower"):
        try:
            r = subprocess.run([SMI, "-d", "0", flag, "--json"], capture_output=True, text=True, timeout=10)
            out["rocm-smi " + flag] = (r.stdout or r.stderr).strip()[:600]
        except Exception as e:
            out["rocm-smi " + flag] = "error: %s" % e
    import glob
    for f in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_cap*")
                    + glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average")):
        try:
            out[f] = open(f).read().strip()
        except Exception as e:
            out[f] = "unreadable: %s" % e
    return out


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
    cap = power_cap()
    print("power cap readout:", json.dumps(cap)[:800], flush=True)
    eng = rtvqa_amd.Engine(0)
    h, w, B = 1080, 1920, 256
    fb = h * w * 3
    from rtvqa_amd.engine import DeviceBuffer, DeviceFrames
    rb, db = DeviceBuffer(eng, fb * (B + 1)), DeviceBuffer(eng, fb * (B + 1))
    for a in range(0, B + 1, 32):
        n = min(32, B + 1 - a)
        r = synth.s_natural(n, h, w, seed=1234, t0=a)
        d = synth.distort(r, t0=a)
        N.check(eng.lib.vqa_copy_h2d(eng.ctx, rb.ptr + a * fb, r.ctypes.data, r.nbytes), "h2d", eng.ctx)
        N.check(eng.lib.vqa_copy_h2d(eng.ctx, db.ptr + a * fb, d.ctypes.data, d.nbytes), "h2d", eng.ctx)
        eng.sync()
    ref, dist = DeviceFrames(rb.ptr, B + 1, h, w, owner=rb), DeviceFrames(db.ptr, B + 1, h, w, owner=db)
    ref_b, dist_b, prev0 = ref.slice(1, B + 1), dist.slice(1, B + 1), dist.frame(0)
    params = eng.make_params(dct_mode=N.DCT_BLOCK8)
    params_fb = eng.make_params(dct_mode=N.DCT_BLOCK8, motion_mode=N.MOTION_FARNEBACK)
    params_full = eng.make_params(dct_mode=N.DCT_FULL)
    dist64 = dist.slice(1, 65)
    planes = bgr_planes(h, w)

    def c3():
        eng.quality_submit(ref_b, dist_b, planes, N.SSIM_GAUSS)
        eng.complexity_submit(dist_b, prev0, N.M_ALL, params)
        eng.quality_wait()
        eng.complexity_wait()

    phases = [
        ("idle", None),
        ("c3 step (all kernels)", c3),
        ("k_ssim_gauss only", lambda: eng.quality(ref_b, dist_b, planes, N.SSIM_GAUSS)),
        ("k_ssim_ffmpeg only", lambda: eng.quality(ref_b, dist_b, planes, N.SSIM_FFMPEG)),
        ("k_block_sad only (+gray)", lambda: eng.complexity(dist_b, prev0=prev0, mask=N.M_MOTION, params=params)),
        ("canny only (+gray)", lambda: eng.complexity(dist_b, prev0=prev0, mask=N.M_EDGE, params=params)),
        ("k_dct8 only (+gray)", lambda: eng.complexity(dist_b, prev0=prev0, mask=N.M_DCT | N.M_TEMPORAL_DCT, params=params)),
        ("gray + histograms only", lambda: eng.complexity(dist_b, prev0=prev0, mask=N.M_GRAY_HIST | N.M_COLOR_HIST, params=params)),
        # the reference-true modes, 64 frames per launch (c3ref's batch)
        ("farneback pyramid only (+gray), 64 pairs", lambda: eng.complexity(dist64, prev0=prev0, mask=N.M_MOTION, params=params_fb)),
        ("full-frame DCT only (+gray), 64 frames", lambda: eng.complexity(dist64, prev0=prev0, mask=N.M_DCT | N.M_TEMPORAL_DCT, params=params_full)),
    ]
    th = threading.Thread(target=poll, daemon=True)
    th.start()
    launches = {}
    for name, fn in phases:
        phase[0] = name
        t0, k = time.time(), 0
        while time.time() - t0 < (2.0 if fn is None else secs):
            if fn is None:
                time.sleep(0.1)
            else:
                fn()
                k += 1
        launches[name] = (k, time.time() - t0)
    stop[0] = True
    th.join(timeout=15)
    summary = {}
    for name, _ in phases:
        s = [x for x in samples if x.get("phase") == name and "sclk_mhz" in x]
        s = s[len(s) // 4:]  # let the governor settle: keep the last three quarters of each phase
        clk = sorted(x["sclk_mhz"] for x in s)
        pw = sorted(x["power_w"] for x in s if "power_w" in x)
        k, dt = launches[name]
        summary[name] = {"samples": len(clk), "sclk_mhz_min": clk[0] if clk else None,
                         "sclk_mhz_median": clk[len(clk) // 2] if clk else None, "sclk_mhz_max": clk[-1] if clk else None,
                         "power_w_median": pw[len(pw) // 2] if pw else None, "iterations": k,
                         "ms_per_iteration": round(dt / k * 1e3, 3) if k else None}
        print("%-28s sclk MHz min/med/max %s/%s/%s  power %s W  %s ms/iter" % (
            name, summary[name]["sclk_mhz_min"], summary[name]["sclk_mhz_median"], summary[name]["sclk_mhz_max"],
            summary[name]["power_w_median"], summary[name]["ms_per_iteration"]), flush=True)
    out = {"tool": "scripts/clock_trace.py", "sampler": SMI + " -d 0 --showclocks --showpower --json, every ~0.1-0.4 s",
           "workload": "256 x 1080p device-resident, synth.s_natural seed 1234", "seconds_per_phase": secs,
           "power_cap_readout": cap, "summary": summary, "samples": [{k: v for k, v in x.items() if k != "raw"} for x in samples],
           "first_raw": next((x.get("raw") for x in samples if x.get("raw")), None)}
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(REPO, "gpurun_out", "clock_trace.json"), "w"), indent=1)
    eng.close()


if __name__ == "__main__":
    main()
