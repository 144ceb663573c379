# (A/B selectors exist in the LAB build only: this script loads csrc/lab/libvqa_hip_lab.so through VQA_LIB_PATH)
export VQA_LIB_PATH=${VQA_LIB_PATH:-$(cd $(dirname $0)/.. && pwd)/real-time-video-quality-analysis_amd/csrc/lab/libvqa_hip_lab.so}
# A/B of the block-SAD kernels on the c3 workload (256 x 1080p): HIP-event ms per launch, natural and noise content
set -o pipefail
mkdir -p gpurun_out
run() { # label content env
  label=$1; content=$2; shift; shift
  env "$@" python3 bench.py --workload c3 --content $content --steps 4 --warmup 1 --cpu-sample 0 --e2e-steps 0 > gpurun_out/sad_ab_$label.json 2> gpurun_out/sad_ab_$label.err || { echo "$label FAILED"; tail -3 gpurun_out/sad_ab_$label.err; return 1; }
  python3 - "$label" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/sad_ab_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
k = d["kernels"]["k_block_sad"]
print("%-22s k_block_sad %.4f ms/launch  frac_hbm %.3f   value %.0f fps" % (sys.argv[1], k["ms_per_launch"], k.get("frac_hbm", 0), d["value"]))
PY
}
run exhaustive_natural natural VQA_SAD_VARIANT=0 && run exhaustive_noise noise VQA_SAD_VARIANT=0 && run pruned_natural natural VQA_SAD_VARIANT=2 && run pruned_noise noise VQA_SAD_VARIANT=2
