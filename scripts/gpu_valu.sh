# VALU occupancy of every kernel of a bench command: rocprofv3 --pmc (no trace domains) with
# SQ_ACTIVE_INST_VALU (QUAD-cycles a SIMD spends issuing VALU: MI355X_MICROARCH.md, "SQ PMC units"), SQ_INSTS_VALU and
# GRBM_GUI_ACTIVE (shader cycles, summed over the 8 XCDs).
# usage: bash scripts/gpu_valu.sh <tag> <bench args...>   -> profiles/<tag>_valu.json
set -o pipefail
tag=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
# one counter per pass (collected together they come back mutually inconsistent on gfx950)
for ctr in SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE; do
  timeout -k 10 600 rocprofv3 --pmc $ctr --output-format csv -d $ROOT/gpurun_out/valu_${tag}_$ctr -- python3 $ROOT/bench.py "$@" --cpu-sample 0 --e2e-steps 0 --api-steps 0 --streams 1 --inflight 1 --no-overlap --no-serial-pass > $ROOT/gpurun_out/valu_${tag}_$ctr.log 2>&1 || { tail -5 $ROOT/gpurun_out/valu_${tag}_$ctr.log; exit 1; }
done
python3 - "$ROOT" "$tag" <<'PY'
import collections, csv, glob, json, os, sys
root, tag = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for ctr in ("SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "GRBM_GUI_ACTIVE"):
    f = max(glob.glob(os.path.join(root, "gpurun_out", "valu_%s_%s" % (tag, ctr), "*", "*counter_collection.csv")), key=os.path.getmtime)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("vqa::", "")
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"tag": tag, "formula": "raw counters per launch: SQ_INSTS_VALU (wave-instructions), GRBM_GUI_ACTIVE / 8 XCDs (shader cycles the launch lasted), SQ_ACTIVE_INST_VALU (quad-cycles).  NOTE: 4 * SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU reads 4.00 for every kernel - the counter counts instructions in quad-cycle units, it is NOT an occupancy; the issue fraction of a launch is computed from SQ_INSTS_VALU and the calibrated per-opcode issue costs (scripts/issue_model.py)", "kernels": {}}
for k, c in sorted(agg.items()):
    if k.startswith("__amd") or "finalize" in k or "finish" in k: continue
    def top(v):  # the full-batch launches (drop the 1-frame prev0 launches)
        big = max(v)
        sel = [x for x in v if x > 0.5 * big]
        return sum(sel) / len(sel), len(sel)
    (g, nl), (a, _), (n, _) = top(c["GRBM_GUI_ACTIVE"]), top(c["SQ_ACTIVE_INST_VALU"]), top(c["SQ_INSTS_VALU"])
    sel = [0] * nl
    out["kernels"][k] = {"launches": len(sel), "valu_insts_per_launch": int(n), "shader_cycles_per_launch": int(g / 8),
                         "sq_active_inst_valu_quadcycles": int(a)}
    print("%-40s %.4g VALU inst/launch  %.4g shader cycles" % (k, n, g / 8))
json.dump(out, open(os.path.join(root, "gpurun_out", "%s_valu.json" % tag), "w"), indent=1)
json.dump(out, open(os.path.join(root, "profiles", "%s_valu.json" % tag), "w"), indent=1)
PY
