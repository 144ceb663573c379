# Round-end artefacts: GPU tests, smoke, PMC passes, then bench lines + rocprofv3 --kernel-trace --stats.
# The PMC passes and their summary come FIRST: bench.py reads roofline.traffic from the newest
# profiles/round<NN>_<workload>_pmc.json whose source hashes match, so the bench lines recorded afterwards carry the
# traffic measured on the same sources.
# usage (from the repo root, on the GPU box): VQA_GIT_SHA=<sha> bash scripts/gpu_round_end.sh [round4]
set -o pipefail
R=${1:-round4}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > $ROOT/gpurun_out/final_pytest.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -3 $ROOT/gpurun_out/final_pytest.log
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 300 python __graft_entry__.py smoke 2>&1 | grep -v amdgpu
bash scripts/gpu_pmc.sh ${R}_c3 --steps 2 --warmup 1 --no-verify || exit 1
bash scripts/gpu_pmc.sh ${R}_c2 --workload c2 --steps 2 --warmup 1 --no-verify || exit 1
bash scripts/gpu_pmc.sh ${R}_c4 --workload c4 --steps 2 --warmup 1 --no-verify || exit 1
python3 scripts/pmc_summarize.py ${R}_c3 256 > /dev/null && python3 scripts/pmc_summarize.py ${R}_c2 256 > /dev/null && python3 scripts/pmc_summarize.py ${R}_c4 64 > /dev/null
cp profiles/${R}_*_pmc.json $ROOT/gpurun_out/ 2>/dev/null
bash scripts/gpu_profile.sh final_c3 --steps 10 --warmup 2 || exit 1
bash scripts/gpu_profile.sh final_c2 --workload c2 --steps 10 --warmup 2 --cpu-sample 32 || exit 1
bash scripts/gpu_profile.sh final_c4 --workload c4 --steps 5 --warmup 2 --cpu-sample 8 || exit 1
bash scripts/gpu_profile.sh final_c3ref --workload c3ref --steps 10 --warmup 2 --cpu-sample 8 || exit 1
bash scripts/gpu_profile.sh final_c3full --dct-mode full --steps 5 --warmup 2 --cpu-sample 0 --e2e-steps 0 || exit 1
bash scripts/gpu_profile.sh final_c2ff --workload c2 --steps 10 --warmup 2 --ssim-mode ffmpeg --cpu-sample 0 --e2e-steps 0 || exit 1
bash scripts/gpu_profile.sh final_c3noise --content noise --steps 5 --warmup 2 --cpu-sample 0 --e2e-steps 0 || exit 1
bash scripts/gpu_profile.sh final_c3fb --motion farneback --batch 64 --steps 10 --warmup 2 --cpu-sample 0 --e2e-steps 0 || exit 1
bash scripts/gpu_valu.sh ${R}_c3 --steps 2 --warmup 1 --no-verify || exit 1
cp profiles/${R}_c3_valu.json $ROOT/gpurun_out/ 2>/dev/null
echo round-end done
