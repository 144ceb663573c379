# usage: bash scripts/gpu_kernel_ms.sh <label> [bench args]: prints every kernel's HIP-event ms per launch for the c3 workload
set -o pipefail
mkdir -p gpurun_out
label=$1; shift
python3 bench.py --workload c3 --steps 5 --warmup 2 --cpu-sample 0 --e2e-steps 0 "$@" > gpurun_out/kms_$label.json 2> gpurun_out/kms_$label.err || { tail -3 gpurun_out/kms_$label.err; exit 1; }
python3 - "$label" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/kms_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], "value %.0f fps  ms/step %.3f" % (d["value"], d["ms_per_step"]), " ".join("%s=%.4f(x%d)" % (k.replace("k_", ""), v["ms_per_launch"], v["launches"]) for k, v in d["kernels"].items()))
PY
