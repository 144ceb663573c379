#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r5
python -m pytest tests/test_gpu_stream.py tests/test_gpu_api.py -x -q -m gpu > gpurun_out/r5/t4.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -15 gpurun_out/r5/t4.log
