set -o pipefail
mkdir -p gpurun_out
python -c "import torch;print(torch.cuda.is_available(), torch.cuda.get_device_name(0))" > gpurun_out/env.log 2>&1
timeout -k 10 900 python -m pytest tests -m gpu -q --maxfail=40 --timeout 180 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -60 gpurun_out/pytest_gpu.log
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; rc=$?; echo "smoke rc=$rc"; tail -5 gpurun_out/smoke.log
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 600 python bench.py --steps 3 --warmup 1 --batch 64 --cpu-sample 8 > gpurun_out/bench_small.log 2>&1; rc=$?; echo "bench rc=$rc"; tail -5 gpurun_out/bench_small.log
