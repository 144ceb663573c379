# Round-end artefacts, in three gpurun calls (each fits the 1200 s limit; then scripts/gpu_api_trace.sh for the entry point's traces):
#   part 1: GPU tests, smoke, the PMC passes (FETCH_SIZE / WRITE_SIZE, VALU counters) and their summaries.  They come FIRST:
#           bench.py reads roofline.traffic from the newest profiles/round<NN>_<workload>_pmc.json whose source hashes match,
#           so the bench lines of part 2 carry the traffic measured on the same sources - COPY the *_pmc.json files of part 1
#           into profiles/ (scripts/collect_profiles.sh) before starting part 2.
#   part 2: smoke + the bench lines + rocprofv3 --kernel-trace --stats of c3 / c1 / c1ref / c2 / c4;
#   part 3: the same for c3ref and the mode variants, then the clock / power trace.
# usage (from the repo root, on the GPU box): VQA_GIT_SHA=<git rev-parse --short HEAD of the COMMITTED tree> bash scripts/gpu_round_end.sh <round tag> <1|2|3>
# (the box has no .git: without VQA_GIT_SHA the PMC files are stamped "uncommitted" - round 5's slip)
set -o pipefail
R=${1:-round6}
PART=${2:-1}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out
if [ "$PART" = "1" ]; then
  timeout -k 10 700 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > $ROOT/gpurun_out/final_pytest.log 2>&1; rc=$?
  echo "pytest rc=$rc"; tail -3 $ROOT/gpurun_out/final_pytest.log
  if [ $rc -ne 0 ]; then exit $rc; fi
  timeout -k 10 300 python __graft_entry__.py smoke 2>&1 | grep -v amdgpu
  bash scripts/gpu_pmc.sh ${R}_c3 --steps 2 --warmup 1 --no-verify || exit 1
  bash scripts/gpu_pmc.sh ${R}_c2 --workload c2 --steps 2 --warmup 1 --no-verify || exit 1
  bash scripts/gpu_pmc.sh ${R}_c4 --workload c4 --steps 2 --warmup 1 --no-verify || exit 1
  python3 scripts/pmc_summarize.py ${R}_c3 256 > /dev/null && python3 scripts/pmc_summarize.py ${R}_c2 256 > /dev/null && python3 scripts/pmc_summarize.py ${R}_c4 64 > /dev/null
  cp profiles/${R}_*_pmc.json $ROOT/gpurun_out/ 2>/dev/null
  bash scripts/gpu_valu.sh ${R}_c3 --steps 2 --warmup 1 --no-verify || exit 1
  cp profiles/${R}_c3_valu.json $ROOT/gpurun_out/ 2>/dev/null
  echo round-end part 1 done
elif [ "$PART" = "2" ]; then
  timeout -k 10 300 python __graft_entry__.py smoke > $ROOT/gpurun_out/${R}_smoke.log 2>&1; echo "smoke rc=$?"; grep -v amdgpu $ROOT/gpurun_out/${R}_smoke.log | tail -2
  bash scripts/gpu_profile.sh final_c3 --steps 10 --warmup 2 || exit 1
  bash scripts/gpu_profile.sh final_c1 --workload c1 --steps 5 --warmup 1 || exit 1
  bash scripts/gpu_profile.sh final_c1ref --workload c1ref --steps 5 --warmup 1 || exit 1
  bash scripts/gpu_profile.sh final_c2 --workload c2 --steps 10 --warmup 2 --cpu-sample 32 || exit 1
  bash scripts/gpu_profile.sh final_c4 --workload c4 --steps 5 --warmup 2 --cpu-sample 8 || exit 1
  echo round-end part 2 done
else
  bash scripts/gpu_profile.sh final_c3ref --workload c3ref --steps 10 --warmup 2 --cpu-sample 8 || exit 1
  bash scripts/gpu_profile.sh final_c3full --dct-mode full --steps 5 --warmup 2 --cpu-sample 0 --e2e-steps 0 || exit 1
  bash scripts/gpu_profile.sh final_c2ff --workload c2 --steps 10 --warmup 2 --ssim-mode ffmpeg --cpu-sample 0 --e2e-steps 0 || exit 1
  bash scripts/gpu_profile.sh final_c3noise --content noise --steps 5 --warmup 2 --cpu-sample 0 --e2e-steps 0 --api-steps 0 || exit 1
  bash scripts/gpu_profile.sh final_c3fb --motion farneback --batch 64 --steps 10 --warmup 2 --cpu-sample 0 --e2e-steps 0 || exit 1
  timeout -k 10 200 python scripts/clock_trace.py 4 > $ROOT/gpurun_out/clock_trace.log 2>&1; echo "clock rc=$?"; tail -12 $ROOT/gpurun_out/clock_trace.log
  echo round-end part 3 done
fi
