# times scripts/dct_full_only.py (64 x 1080p, or h w) for each library given
cd /tmp && export TMPDIR=/tmp
HW="$1 $2"; shift; shift
for lib in "$@"; do
  export VQA_LIB_PATH=$GRAFT_REPO_ROOT/$lib
  d=$GRAFT_REPO_ROOT/gpurun_out/prof_dctab_$(basename $lib .so)
  rm -rf $d
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $GRAFT_REPO_ROOT/scripts/dct_full_only.py $HW > $d.log 2>&1 || { tail -5 $d.log; exit 1; }
  echo "== $lib ($HW)"
  f=$(ls -t $d/*/*kernel_stats.csv | head -1)
  python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'dct_fft' in r["Name"]: print("%-44s calls %3s avg %9.1f us"%(r["Name"].split("(")[0].replace("void vqa::","")[:44], r["Calls"], float(r["AverageNs"])/1e3))
PY
done
