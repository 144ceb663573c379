cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_fb2 -- python3 $GRAFT_REPO_ROOT/scripts/fb_only.py > $GRAFT_REPO_ROOT/gpurun_out/fb_prof2.log 2>&1
grep farneback $GRAFT_REPO_ROOT/gpurun_out/fb_prof2.log
f=$(ls -t $GRAFT_REPO_ROOT/gpurun_out/prof_fb2/*/*kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'fb' in r["Name"]: print("%-40s calls %4s total %8.3f ms avg %8.1f us  %5s%%"%(r["Name"].split("(")[0][-40:], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3, r["Percentage"]))
PY
