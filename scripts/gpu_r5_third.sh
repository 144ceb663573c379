#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r5
python -m pytest tests/test_gpu_api.py tests/test_gpu_stream.py tests/test_golden_pipeline.py tests/test_skimage_pins.py -x -q -m gpu > gpurun_out/r5/t3.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 gpurun_out/r5/t3.log
[ $rc -eq 0 ] || exit $rc
python bench.py --workload c1 --steps 5 > gpurun_out/r5/bench_c1b.json 2> gpurun_out/r5/bench_c1b.err; echo "bench c1 rc=$?"; python - <<'PY'
import json
l=json.load(open('gpurun_out/r5/bench_c1b.json'))
print(l["value"], l["cpu_baseline"], l["api_end_to_end"], l["verified"]["ok"])
PY
