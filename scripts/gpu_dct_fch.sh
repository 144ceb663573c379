# (A/B selectors exist in the LAB build only: this script loads csrc/lab/libvqa_hip_lab.so through VQA_LIB_PATH)
export VQA_LIB_PATH=${VQA_LIB_PATH:-$(cd $(dirname $0)/.. && pwd)/real-time-video-quality-analysis_amd/csrc/lab/libvqa_hip_lab.so}
set -o pipefail
mkdir -p gpurun_out
for wl in c2 c4; do
for f in 4 8 12 16 20 24 32 43; do
  VQA_DCT_FCH=$f python3 bench.py --workload $wl --steps 4 --warmup 1 --cpu-sample 0 --e2e-steps 0 > gpurun_out/fch.json 2> gpurun_out/fch.err || { tail -3 gpurun_out/fch.err; exit 1; }
  python3 -c "
import json
d=json.loads(open('gpurun_out/fch.json').read().strip().splitlines()[-1])
print('$wl fch=$f  k_dct8 %.4f ms' % d['kernels']['k_dct8']['ms_per_launch'])"
done; done
