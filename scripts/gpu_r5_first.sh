#!/bin/bash
# round 5, first GPU pass: the new host side (stream.py), vqa_trim / table caches, then the API rates
set -o pipefail
mkdir -p gpurun_out/r5
python -m pytest tests/test_gpu_stream.py tests/test_gpu_api.py tests/test_abi.py -x -q -m gpu > gpurun_out/r5/t1.log 2>&1; echo "t1 rc=$?" 
tail -5 gpurun_out/r5/t1.log
python -m pytest tests/test_golden_pipeline.py -x -q -m gpu > gpurun_out/r5/t2.log 2>&1; echo "t2 rc=$?"
tail -3 gpurun_out/r5/t2.log
timeout -k 10 400 python scripts/api_rate.py 257 > gpurun_out/r5/api_rate.log 2>&1; echo "api rc=$?"
cat gpurun_out/r5/api_rate.log
