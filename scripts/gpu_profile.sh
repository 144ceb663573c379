# usage: bash scripts/gpu_profile.sh <tag> <bench args...>
# 1. bench.py with the given arguments, plainly: the HEADLINE line of that workload (gpurun_out/bench_<tag>.log).
# 2. the SERIAL configuration of the same workload (one context, overlap off, one batch in flight - what the headline
#    run's serial pass measures) under rocprofv3 --kernel-trace --stats: every launch of that run is serial, so the
#    kernel averages of gpurun_out/prof_<tag>/ agree with the HIP-event `kernels` table of both bench lines; its own
#    bench line is kept as gpurun_out/bench_<tag>_serial.log.
set -o pipefail
tag=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out
timeout -k 10 900 python bench.py "$@" > $ROOT/gpurun_out/bench_$tag.log 2> $ROOT/gpurun_out/bench_$tag.err; rc=$?
echo "bench rc=$rc"; tail -2 $ROOT/gpurun_out/bench_$tag.err; tail -1 $ROOT/gpurun_out/bench_$tag.log | cut -c1-200
if [ $rc -ne 0 ]; then exit $rc; fi
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_$tag -- python3 $ROOT/bench.py "$@" --cpu-sample 0 --e2e-steps 0 --api-steps 0 --streams 1 --inflight 1 --no-overlap > $ROOT/gpurun_out/bench_${tag}_serial.log 2> $ROOT/gpurun_out/rocprof_$tag.err; rc=$?
echo "rocprof rc=$rc"; tail -2 $ROOT/gpurun_out/rocprof_$tag.err
find $ROOT/gpurun_out/prof_$tag -name "*kernel_stats*" | head -3
exit $rc
