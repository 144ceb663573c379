# (A/B selectors exist in the LAB build only: this script loads csrc/lab/libvqa_hip_lab.so through VQA_LIB_PATH)
export VQA_LIB_PATH=${VQA_LIB_PATH:-$(cd $(dirname $0)/.. && pwd)/real-time-video-quality-analysis_amd/csrc/lab/libvqa_hip_lab.so}
# usage: bash scripts/gpu_env_ab.sh "<bench args>" VAR=a VAR=b ...   (c3 per-kernel ms for each environment setting)
args=$1; shift
for v in "$@"; do env $v bash scripts/gpu_kernel_ms.sh $(echo $v | tr '=' '_') $args || exit 1; done
