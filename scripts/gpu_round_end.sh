# Round-end artefacts: GPU tests, smoke, bench lines + rocprofv3 --kernel-trace --stats + PMC passes (separate runs).
# usage (from the repo root, on the GPU box): VQA_GIT_SHA=<sha> bash scripts/gpu_round_end.sh
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > $ROOT/gpurun_out/final_pytest.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -3 $ROOT/gpurun_out/final_pytest.log
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 300 python __graft_entry__.py smoke 2>&1 | grep -v amdgpu
bash scripts/gpu_profile.sh final_c3 --steps 10 --warmup 2 || exit 1
bash scripts/gpu_profile.sh final_c2 --workload c2 --steps 10 --warmup 2 --cpu-sample 32 || exit 1
bash scripts/gpu_profile.sh final_c4 --workload c4 --steps 5 --warmup 2 --cpu-sample 8 || exit 1
bash scripts/gpu_profile.sh final_c2ff --workload c2 --steps 10 --warmup 2 --ssim-mode ffmpeg --cpu-sample 0 --e2e-steps 0 || exit 1
bash scripts/gpu_profile.sh final_c3noise --content noise --steps 5 --warmup 2 --cpu-sample 0 --e2e-steps 0 || exit 1
bash scripts/gpu_profile.sh final_c3fb --motion farneback --batch 64 --steps 3 --warmup 1 --cpu-sample 0 --e2e-steps 0 || exit 1
bash scripts/gpu_pmc.sh final_c3 --steps 2 --warmup 1 || exit 1
bash scripts/gpu_pmc.sh final_c2 --workload c2 --steps 2 --warmup 1 || exit 1
bash scripts/gpu_pmc.sh final_c4 --workload c4 --steps 2 --warmup 1 || exit 1
bash scripts/gpu_valu.sh final_c3 --steps 2 --warmup 1 || exit 1
python3 scripts/pmc_summarize.py final_c3 256 > /dev/null && python3 scripts/pmc_summarize.py final_c2 256 > /dev/null && python3 scripts/pmc_summarize.py final_c4 64 > /dev/null
cp profiles/final_*_pmc.json profiles/final_*_valu.json $ROOT/gpurun_out/ 2>/dev/null
echo round-end done
