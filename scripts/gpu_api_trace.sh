# The reference-shaped entry point under rocprofv3 with the pass's roctx ranges (SURVEY.md section 5 "Tracing"): one run per
# residence of the clip (kernel + memory-copy + marker traces, no counters), each folded by scripts/trace_summary.py into
# gpurun_out/<round>_api_trace_<residence>.json: H2D busy time, kernel busy time, their overlap, the idle gaps, the per-lane
# turnaround.  Then the same driver once WITHOUT the profiler (the rates the traced runs should be read against).
# usage (repo root, on the GPU box): VQA_GIT_SHA=<sha> bash scripts/gpu_api_trace.sh <round tag> [frames, default 257]
set -o pipefail
R=${1:-round6}
NF=${2:-257}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for res in pinned pageable resident; do
  rm -rf $ROOT/gpurun_out/api_trace_$res
  timeout -k 10 400 rocprofv3 --kernel-trace --memory-copy-trace --marker-trace --stats --output-format csv -d $ROOT/gpurun_out/api_trace_$res -- python3 $ROOT/scripts/api_trace.py $NF $res 3 > $ROOT/gpurun_out/api_trace_$res.log 2>&1; rc=$?
  echo "rocprofv3 $res rc=$rc"; grep -E "^(resident|host_|roctx)" $ROOT/gpurun_out/api_trace_$res.log
  if [ $rc -ne 0 ]; then tail -5 $ROOT/gpurun_out/api_trace_$res.log; exit $rc; fi
  python3 $ROOT/scripts/trace_summary.py $ROOT/gpurun_out/api_trace_$res $res $ROOT/gpurun_out/api_trace_$res.log > $ROOT/gpurun_out/${R}_api_trace_$res.json || exit 1
  f=$(find $ROOT/gpurun_out/api_trace_$res -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" $ROOT/gpurun_out/${R}_api_trace_${res}_kernel_stats.csv
  python3 - $ROOT/gpurun_out/${R}_api_trace_$res.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print({k: d.get(k) for k in ("window_ms", "kernel_busy_ms", "h2d_busy_ms", "kernel_and_h2d_overlap_ms", "neither_ms", "overlap_frac_of_h2d", "idle_frac", "lane_turnaround_ms")})
PY
  # the raw traces are large: keep the summaries
  find $ROOT/gpurun_out/api_trace_$res -name "*trace.csv" -size +8M -delete
done
VQA_ROCTX=0 timeout -k 10 300 python3 $ROOT/scripts/api_trace.py $NF all 3 > $ROOT/gpurun_out/${R}_api_untraced.log 2>&1; echo "untraced rc=$?"; cat $ROOT/gpurun_out/${R}_api_untraced.log | grep -v amdgpu
