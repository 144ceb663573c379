# usage: bash scripts/gpu_env_ab.sh "<bench args>" VAR=a VAR=b ...   (c3 per-kernel ms for each environment setting)
args=$1; shift
for v in "$@"; do env $v bash scripts/gpu_kernel_ms.sh $(echo $v | tr '=' '_') $args || exit 1; done
