#!/usr/bin/env python3
"""Fold the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of scripts/gpu_pmc.sh into
profiles/<tag>_pmc.json: per kernel, mean counter values per launch and the HBM bytes after the
guide's gfx950 correction (FETCH_SIZE is in KiB and counts 128-B requests as 64 B: double it;
WRITE_SIZE in KiB is exact).  usage: pmc_summarize.py <tag> <frames_per_launch>"""
import collections, csv, glob, hashlib, json, os, subprocess, sys
tag, frames = sys.argv[1], int(sys.argv[2])
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csrc = os.path.join(root, "real-time-video-quality-analysis_amd", "csrc")
# bench.py uses a PMC file only while these hashes match the sources it runs on (roofline.traffic_source)
hashes = {os.path.basename(f): hashlib.sha256(open(f, "rb").read()).hexdigest()[:16]
          for f in sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.hpp")))}
sha = os.environ.get("VQA_GIT_SHA", "")
if not sha:
    try:
        sha = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    except Exception:
        sha = ""
out = {"tag": tag, "frames_per_launch": frames, "unit": "bytes per launch", "git_sha": sha or "uncommitted", "source_sha256": hashes,
       "correction": "hbm_bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (MI355X_MICROARCH.md §HBM)", "kernels": {}}
vals = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    f = max(glob.glob(os.path.join(root, "gpurun_out", "pmc_%s_%s" % (tag, ctr), "*", "*counter_collection.csv")),
            key=os.path.getmtime)  # gpurun_out accumulates: take the newest pass
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("vqa::", "")
        agg[name].append(float(r["Counter_Value"]))
    vals[ctr] = agg
for name in vals["FETCH_SIZE"]:
    if name.startswith("__amd") or "finalize" in name or "finish" in name:
        continue
    fs, ws = vals["FETCH_SIZE"][name], vals["WRITE_SIZE"].get(name, [0.0])
    big = max(fs)
    fsel = [v for v in fs if v > 0.5 * big]  # drop the 1-frame prev0 launches
    wbig = max(ws)
    wsel = [v for v in ws if v > 0.5 * wbig] or [0.0]
    fetch, write = sum(fsel) / len(fsel), sum(wsel) / len(wsel)
    out["kernels"][name] = {"FETCH_SIZE_KiB": round(fetch, 1), "WRITE_SIZE_KiB": round(write, 1), "launches": len(fsel),
                            "hbm_bytes": int(2 * fetch * 1024 + write * 1024)}
path = os.path.join(root, "profiles", "%s_pmc.json" % tag)
json.dump(out, open(path, "w"), indent=1)
print(json.dumps(out, indent=1))
