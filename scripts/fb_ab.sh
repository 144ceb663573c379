# Farneback A/B on the GPU box: the parity tests of the mode, then the pyramid alone (32 x 1080p pairs) under rocprofv3 -
# shipped (fused iteration) and, with the lab library, the two-kernel form of rounds 2-3 (VQA_FB_VARIANT=1).
set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests -q -x -m gpu -k "farneback or golden_pipeline or aggregator or c3ref" > gpurun_out/fb_tests.log 2>&1 || { tail -30 gpurun_out/fb_tests.log; exit 1; }
tail -2 gpurun_out/fb_tests.log
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  if [ $v = 1 ]; then export VQA_LIB_PATH=$GRAFT_REPO_ROOT/real-time-video-quality-analysis_amd/csrc/lab/libvqa_hip_lab.so VQA_FB_VARIANT=1; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_fbab$v -- python3 $GRAFT_REPO_ROOT/scripts/fb_only.py > $GRAFT_REPO_ROOT/gpurun_out/fb_ab$v.log 2>&1
  grep farneback $GRAFT_REPO_ROOT/gpurun_out/fb_ab$v.log
  f=$(ls -t $GRAFT_REPO_ROOT/gpurun_out/prof_fbab$v/*/*kernel_stats.csv | head -1)
  python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'fb' in r["Name"]: print("%-40s calls %4s total %8.3f ms avg %8.1f us  %5s%%"%(r["Name"].split("(")[0][-40:], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3, r["Percentage"]))
PY
done
