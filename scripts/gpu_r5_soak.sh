#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r5
timeout -k 10 500 python scripts/fuzz_stream.py 600 5100000 > gpurun_out/r5/fuzz_stream.log 2>&1; rc=$?; echo "fuzz_stream rc=$rc"; tail -3 gpurun_out/r5/fuzz_stream.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python scripts/fuzz_parity.py 2000 5200000 > gpurun_out/r5/fuzz_parity.log 2>&1; rc=$?; echo "fuzz_parity rc=$rc"; tail -3 gpurun_out/r5/fuzz_parity.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python scripts/fuzz_farneback.py 400 5300000 > gpurun_out/r5/fuzz_fb.log 2>&1; rc=$?; echo "fuzz_fb rc=$rc"; tail -4 gpurun_out/r5/fuzz_fb.log
