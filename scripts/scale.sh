#!/bin/bash
# The scaling curve BASELINE.json asks for (frames/sec at 1, 2, 4, 8 GPUs of ONE node), exactly as the driver launches
# bench.py.  Weak scaling: one synthetic stream per GPU, no data-path collective, one scalar RCCL all-reduce after the
# timed region.  Needs a multi-GPU node; NO run of this repository has had one (the builder's boxes have one GPU), so no
# scaling number exists yet.  usage:  bash scripts/scale.sh [workload: c3 (default, the headline) | c4 = BASELINE configs[4]]
# Each line of gpurun_out/scale_<workload>.jsonl is one bench line; efficiency = value(N) / (N * value(1)).
set -o pipefail
WL=${1:-c3}
STEPS=${STEPS:-10}
WARMUP=${WARMUP:-2}
PORT=${PORT:-29511}
export HSA_ENABLE_IPC_MODE_LEGACY=0   # dmabuf IPC: RCCL needs it on this pool
mkdir -p gpurun_out
OUT=gpurun_out/scale_$WL.jsonl
: > $OUT
# N = 1: plain python (no launcher); the CPU baseline leg runs only here
python bench.py --gpus 1 --workload $WL --steps $STEPS --warmup $WARMUP | tee -a $OUT || exit 1
for N in 2 4 8; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $PORT \
      bench.py --gpus $N --workload $WL --steps $STEPS --warmup $WARMUP | tee -a $OUT || exit 1
  PORT=$((PORT + 1))
done
python3 - "$OUT" <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
one = rows[0]["value"]
for r in rows:
    print("N=%d  %10.1f frames/s  efficiency %.3f  devices %s  collective %s" % (
        r["n_gpus"], r["value"], r["value"] / (r["n_gpus"] * one), r["config"]["devices"], r["config"]["collective"]))
PY
# the same configuration without Python (one process, one host thread + context per device, ncclCommInitAll):
#   gcc -O2 -pthread -Iinclude -o /tmp/vqa_multi examples/vqa_multi.c -Lreal-time-video-quality-analysis_amd/csrc -lvqa_hip \
#       -Wl,-rpath,$PWD/real-time-video-quality-analysis_amd/csrc -lm && /tmp/vqa_multi 8 64 5
