# usage: bash scripts/gpu_pmc.sh <tag> <bench args...>
# HBM traffic counters for the bench command, in SEPARATE passes (FETCH_SIZE and WRITE_SIZE do not fit
# one pass on gfx950) and WITHOUT any trace domain (gpurun refuses --pmc together with a trace).  The command is the
# bench's SERIAL configuration (one context, overlap off): counters per launch, not perturbed by a neighbour kernel.
set -o pipefail
tag=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 600 rocprofv3 --pmc $ctr --output-format csv -d $ROOT/gpurun_out/pmc_${tag}_$ctr -- python3 $ROOT/bench.py "$@" --cpu-sample 0 --e2e-steps 0 --api-steps 0 --streams 1 --inflight 1 --no-overlap --no-serial-pass > $ROOT/gpurun_out/pmc_${tag}_$ctr.log 2>&1; rc=$?
  echo "$ctr rc=$rc"; tail -2 $ROOT/gpurun_out/pmc_${tag}_$ctr.log | cut -c1-300
  if [ $rc -ge 124 ]; then exit $rc; fi
done
find $ROOT/gpurun_out/pmc_${tag}_* -name "*counter_collection*" | head
