# (A/B selectors exist in the LAB build only: this script loads csrc/lab/libvqa_hip_lab.so through VQA_LIB_PATH)
export VQA_LIB_PATH=${VQA_LIB_PATH:-$(cd $(dirname $0)/.. && pwd)/real-time-video-quality-analysis_amd/csrc/lab/libvqa_hip_lab.so}
for wl in c4 c3; do for W in 4 8 12 16; do
  export VQA_HYST_WIDE=$W
  timeout -k 10 200 python bench.py --workload $wl --steps 3 --warmup 1 --cpu-sample 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$wl WIDE=$W', d['value'], d['ms_per_step'], d['kernels']['k_canny_hyst']['ms_per_launch'])
" || exit 1
done; done
