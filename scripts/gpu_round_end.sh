# Round-end artefacts: GPU tests, smoke, default bench + rocprofv3 stats + PMC passes (c2 default, c3).
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider > $ROOT/gpurun_out/final_pytest.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -3 $ROOT/gpurun_out/final_pytest.log
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 300 python __graft_entry__.py smoke 2>&1 | grep -v amdgpu
bash scripts/gpu_profile.sh final_c2 --steps 10 --warmup 2 || exit 1
bash scripts/gpu_profile.sh final_c3 --workload c3 --steps 5 --warmup 2 --cpu-sample 16 || exit 1
bash scripts/gpu_profile.sh final_c2ff --steps 10 --warmup 2 --ssim-mode ffmpeg --cpu-sample 0 || exit 1
bash scripts/gpu_profile.sh final_c4 --workload c4 --steps 5 --warmup 2 --cpu-sample 4 || exit 1
bash scripts/gpu_profile.sh final_c3fb --workload c3 --motion farneback --batch 64 --steps 3 --warmup 1 --cpu-sample 0 || exit 1
bash scripts/gpu_pmc.sh final_c2 --steps 2 --warmup 1 || exit 1
bash scripts/gpu_pmc.sh final_c3 --workload c3 --steps 2 --warmup 1 || exit 1
bash scripts/gpu_pmc.sh final_c4 --workload c4 --steps 2 --warmup 1 || exit 1
