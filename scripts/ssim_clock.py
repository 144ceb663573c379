"""What costs k_ssim_gauss its clock?  

For each build of libvqa_hip.so given on the command line (label=path; the shipped library is always first), one child
process loads it through VQA_LIB_PATH, keeps k_ssim_gauss alone queued on a 256 x 1080p device-resident batch (B, G, R
planes: the c3 launch) for a few seconds, samples `rocm-smi --showclocks --showpower` meanwhile, and times the launch
with the library's own HIP events.  Reported per build: median shader clock, median socket power, ms per launch, and
shader cycles per launch = ms x clock (the quantity that says whether a variant does less work or merely runs at
another frequency).  Output: gpurun_out/ssim_clock.json (copy to profiles/round4_ssim_clock.json).

usage (GPU box): python scripts/ssim_clock.py [seconds] shipped=<path> noLDSreads=gpurun_ab/libvqa_SSIM_PROBE_1.so ...
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
SMI = "/opt/rocm/bin/rocm-smi"


def child(secs):
    sys.path.insert(0, REPO)
    import rtvqa_amd
    from rtvqa_amd import _native as N, synth
    from rtvqa_amd.engine import DeviceBuffer, DeviceFrames, bgr_planes
    samples, stop = [], [False]

    def poll():
        while not stop[0]:
            try:
                out = subprocess.run([SMI, "-d", "0", "--showclocks", "--showpower", "--json"], capture_output=True, text=True,
                                     timeout=10).stdout
                rec = {"t": time.time()}
                card = next(iter(json.loads(out).values()))
                for k, v in card.items():
                    m = re.search(r"\((\d+)Mhz\)", str(v))
                    if k.lower().startswith("sclk") and m:
                        rec["sclk_mhz"] = int(m.group(1))
                    elif "power" in k.lower() and "(w)" in k.lower():
                        try:
                            rec["power_w"] = float(v)
                        except ValueError:
                            pass
                samples.append(rec)
            except Exception:
                pass
            time.sleep(0.05)

    eng = rtvqa_amd.Engine(0)
    h, w, B = 1080, 1920, 256
    fb = h * w * 3
    rb, db = DeviceBuffer(eng, fb * B), DeviceBuffer(eng, fb * B)
    for a in range(0, B, 32):
        r = synth.s_natural(32, h, w, seed=1234, t0=a)
        d = synth.distort(r, t0=a)
        N.check(eng.lib.vqa_copy_h2d(eng.ctx, rb.ptr + a * fb, r.ctypes.data, r.nbytes), "h2d", eng.ctx)
        N.check(eng.lib.vqa_copy_h2d(eng.ctx, db.ptr + a * fb, d.ctypes.data, d.nbytes), "h2d", eng.ctx)
        eng.sync()
    ref, dist = DeviceFrames(rb.ptr, B, h, w, owner=rb), DeviceFrames(db.ptr, B, h, w, owner=db)
    planes = bgr_planes(h, w)
    for _ in range(3):
        q = eng.quality(ref, dist, planes, N.SSIM_GAUSS)
    th = threading.Thread(target=poll, daemon=True)
    th.start()
    eng.profile(True)
    eng.profile_read(reset=True)
    t0, k = time.time(), 0
    while time.time() - t0 < secs:
        eng.quality(ref, dist, planes, N.SSIM_GAUSS)
        k += 1
    prof = eng.profile_read(reset=True)
    stop[0] = True
    th.join(timeout=15)
    s = [x for x in samples if "sclk_mhz" in x]
    s = s[len(s) // 4:]  # let the governor settle
    clk = sorted(x["sclk_mhz"] for x in s)
    pw = sorted(x["power_w"] for x in s if "power_w" in x)
    ms, cnt = prof["k_ssim_gauss"]
    med = clk[len(clk) // 2] if clk else None
    out = {"flavour": eng.lib.vqa_build_flavour(), "launches": cnt, "ms_per_launch": round(ms / cnt, 4),
           "sclk_mhz_min": clk[0] if clk else None, "sclk_mhz_median": med, "sclk_mhz_max": clk[-1] if clk else None,
           "power_w_median": pw[len(pw) // 2] if pw else None, "samples": len(clk),
           "mcycles_per_launch": round(ms / cnt * 1e-3 * med, 3) if med else None,
           "ssim_b_frame0": float(q[0, 0]["ssim"]), "sse_b_frame0": int(q[0, 0]["sse"])}
    print("RESULT " + json.dumps(out), flush=True)
    eng.close()


def main():
    args = sys.argv[1:]
    if args and args[0] == "--child":
        return child(float(args[1]))
    secs = 5.0
    if args and re.match(r"^[0-9.]+$", args[0]):
        secs = float(args.pop(0))
    builds = [a.split("=", 1) for a in args] or [["shipped", ""]]
    res = {}
    for label, path in builds:
        env = dict(os.environ)
        env.pop("VQA_LIB_PATH", None)
        if path:
            env["VQA_LIB_PATH"] = os.path.abspath(path)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(secs)], env=env, capture_output=True,
                           text=True, timeout=600)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        if r.returncode or not line:
            print(label, "FAILED", r.stderr[-600:], flush=True)
            continue
        res[label] = dict(json.loads(line[0][7:]), lib=path or "csrc/libvqa_hip.so")
        d = res[label]
        print("%-22s %8.4f ms  sclk %s MHz (min %s max %s)  %s W  %.3f Mcycles  ssim %.6f" % (
            label, d["ms_per_launch"], d["sclk_mhz_median"], d["sclk_mhz_min"], d["sclk_mhz_max"], d["power_w_median"],
            d["mcycles_per_launch"] or 0, d["ssim_b_frame0"]), flush=True)
        time.sleep(2.0)  # let the chip cool to the same starting point
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    json.dump({"tool": "scripts/ssim_clock.py", "workload": "k_ssim_gauss alone, 256 x 1080p B/G/R planes per launch, device-resident",
               "seconds_per_build": secs, "sampler": SMI + " -d 0 --showclocks --showpower --json", "builds": res},
              open(os.path.join(REPO, "gpurun_out", "ssim_clock.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
