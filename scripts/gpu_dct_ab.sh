# (A/B selectors exist in the LAB build only: this script loads csrc/lab/libvqa_hip_lab.so through VQA_LIB_PATH)
export VQA_LIB_PATH=${VQA_LIB_PATH:-$(cd $(dirname $0)/.. && pwd)/real-time-video-quality-analysis_amd/csrc/lab/libvqa_hip_lab.so}
# A/B of the 8x8 DCT kernel variants on the c2 workload (256 x 1080p): prints k_dct8's HIP-event ms per launch
set -o pipefail
mkdir -p gpurun_out
run() { # label, env...
  label=$1; shift
  env "$@" python3 bench.py --workload c2 --steps 5 --warmup 2 --cpu-sample 0 --e2e-steps 0 > gpurun_out/dct_ab_$label.json 2> gpurun_out/dct_ab_$label.err || { echo "$label FAILED"; tail -3 gpurun_out/dct_ab_$label.err; return 1; }
  python3 - "$label" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/dct_ab_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
k = d["kernels"]["k_dct8"]
print("%-14s k_dct8 %.4f ms/launch  frac_hbm %.3f   value %.0f fps" % (sys.argv[1], k["ms_per_launch"], k.get("frac_hbm", 0), d["value"]))
PY
}
for v in "$@"; do run $(echo $v | tr '=' '_') $v || exit 1; done
