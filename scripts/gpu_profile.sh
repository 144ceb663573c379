# usage: bash scripts/gpu_profile.sh <tag> <bench args...>
# Runs bench.py once plainly and once under rocprofv3 --kernel-trace --stats; leaves the
# stats CSVs under gpurun_out/prof_<tag>/ and the bench JSON lines in gpurun_out/.
set -o pipefail
tag=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out
timeout -k 10 600 python bench.py "$@" > $ROOT/gpurun_out/bench_$tag.log 2>&1; rc=$?
echo "bench rc=$rc"; tail -2 $ROOT/gpurun_out/bench_$tag.log
if [ $rc -ge 124 ]; then exit $rc; fi
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_$tag -- python3 $ROOT/bench.py "$@" --cpu-sample 0 --e2e-steps 0 > $ROOT/gpurun_out/rocprof_$tag.log 2>&1; rc=$?
echo "rocprof rc=$rc"; tail -3 $ROOT/gpurun_out/rocprof_$tag.log
find $ROOT/gpurun_out/prof_$tag -name "*kernel_stats*" | head
