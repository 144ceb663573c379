# HBM traffic counters of the Farneback kernels (scripts/fb_only.py), separate passes, no trace domains.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $ctr --output-format csv -d $ROOT/gpurun_out/pmc_fb_$ctr -- python3 $ROOT/scripts/fb_only.py > $ROOT/gpurun_out/pmc_fb_$ctr.log 2>&1 || exit 1
done
python3 - "$ROOT" <<'PY'
import csv, glob, os, sys, collections
root = sys.argv[1]
vals = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    f = max(glob.glob(os.path.join(root, "gpurun_out", "pmc_fb_%s" % ctr, "*", "*counter_collection.csv")), key=os.path.getmtime)
    agg = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0].replace("void ", "").replace("vqa::", "")] += float(r["Counter_Value"])
    vals[ctr] = agg
P = 1080 * 1920 * 32 * 3  # pixel-pairs processed by the three launches of 32 pairs
tot = 0
for k in sorted(vals["FETCH_SIZE"]):
    if not k.startswith("k_fb"): continue
    b = 2 * vals["FETCH_SIZE"][k] * 1024 + vals["WRITE_SIZE"].get(k, 0) * 1024
    tot += b
    print("%-22s %8.1f B per level-0 pixel-pair (fetch %.1f, write %.1f)" % (k, b / P, 2 * vals["FETCH_SIZE"][k] * 1024 / P, vals["WRITE_SIZE"].get(k, 0) * 1024 / P))
print("total %.1f B per level-0 pixel-pair" % (tot / P))
import json, subprocess
out = {"what": "HBM bytes of the Farneback kernels, scripts/fb_only.py (3 launches of 32 pairs of 1080p), per level-0 pixel and pair, "
               "summed over the four pyramid levels; hbm = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (MI355X_MICROARCH.md, HBM section)",
       "git_sha": os.environ.get("VQA_GIT_SHA", ""), "pixel_pairs": P,
       "kernels": {k: {"fetch_B_per_px_pair": round(2 * vals["FETCH_SIZE"][k] * 1024 / P, 2),
                       "write_B_per_px_pair": round(vals["WRITE_SIZE"].get(k, 0) * 1024 / P, 2)}
                   for k in sorted(vals["FETCH_SIZE"]) if k.startswith("k_fb")},
       "total_B_per_px_pair": round(tot / P, 1)}
json.dump(out, open(os.path.join(root, "gpurun_out", "fb_pmc.json"), "w"), indent=1)
PY
