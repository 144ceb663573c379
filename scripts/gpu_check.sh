# Quick GPU check during development: GPU tests, smoke, one default bench line.
# usage (GPU box, repo root): bash scripts/gpu_check.sh [pytest -k expression]
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out
if [ -n "$1" ]; then K=(-k "$1"); else K=(); fi
timeout -k 10 1000 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -x "${K[@]}" > $ROOT/gpurun_out/check_pytest.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -25 $ROOT/gpurun_out/check_pytest.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python __graft_entry__.py smoke 2>&1 | grep -v amdgpu
timeout -k 10 600 python bench.py > $ROOT/gpurun_out/check_bench.json 2> $ROOT/gpurun_out/check_bench.err; rc=$?
echo "bench rc=$rc"; tail -3 $ROOT/gpurun_out/check_bench.err; cut -c1-1500 $ROOT/gpurun_out/check_bench.json
