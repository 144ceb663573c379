#!/bin/bash
# round 5, second GPU pass: the whole GPU suite, the default bench line (api_end_to_end), c1, the resident API profile
set -o pipefail
mkdir -p gpurun_out/r5
python -m pytest tests -x -q -m gpu > gpurun_out/r5/gpu_tests.log 2>&1; rc=$?; echo "gpu tests rc=$rc"; tail -4 gpurun_out/r5/gpu_tests.log
[ $rc -eq 0 ] || exit $rc
python bench.py > gpurun_out/r5/bench_c3.json 2> gpurun_out/r5/bench_c3.err; echo "bench c3 rc=$?"; tail -c 1500 gpurun_out/r5/bench_c3.json
python bench.py --workload c1 --steps 5 > gpurun_out/r5/bench_c1.json 2> gpurun_out/r5/bench_c1.err; echo "bench c1 rc=$?"; tail -c 3000 gpurun_out/r5/bench_c1.json; tail -5 gpurun_out/r5/bench_c1.err
API_PROFILE=1 timeout -k 10 300 python scripts/api_rate.py 257 > gpurun_out/r5/api_rate2.log 2>&1; echo "api rc=$?"
