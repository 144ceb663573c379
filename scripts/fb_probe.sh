# times scripts/fb_only.py under rocprofv3 for each library given (paths relative to the repo root)
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  export VQA_LIB_PATH=$GRAFT_REPO_ROOT/$lib
  d=$GRAFT_REPO_ROOT/gpurun_out/prof_fbp_$(basename $lib .so)
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $GRAFT_REPO_ROOT/scripts/fb_only.py > $d.log 2>&1 || { tail -5 $d.log; exit 1; }
  echo "== $lib"; grep farneback $d.log | tail -1
  f=$(ls -t $d/*/*kernel_stats.csv | head -1)
  python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'fb_iter' in r["Name"]: print("%-40s calls %4s total %8.3f ms max %8.1f us"%(r["Name"].split("(")[0][-40:], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["MaxNs"])/1e3))
PY
done
