# rocprofv3 kernel stats of the full-frame DCT path alone (64 x 1080p, or the size given): bash scripts/dct_fft_profile.sh [h w]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/prof_dctfft
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_dctfft -- python3 $ROOT/scripts/dct_full_only.py "$@" > $ROOT/gpurun_out/dctfft.log 2>&1
f=$(ls -t $ROOT/gpurun_out/prof_dctfft/*/*kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].split("(")[0].replace("void ", "").replace("vqa::", "")[:40]
    if "dct" in n or "final" in n or "gemm" in n:
        print("%-40s calls %3s avg %9.1f us" % (n, r["Calls"], float(r["AverageNs"]) / 1e3))
PY
