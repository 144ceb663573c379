# After `gpurun -- bash scripts/gpu_round_end.sh`: copy the judged summaries from gpurun_out/ (scratch) into
# profiles/ (tracked).  usage: bash scripts/collect_profiles.sh [round_tag, default round1]
set -e
R=${1:-round1}
cd "$(dirname "$0")/.."
for t in c2 c3 c2ff c4 c3fb; do
  f=$(ls -t gpurun_out/prof_final_$t/*/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && cp "$f" profiles/${R}_${t}_kernel_stats.csv
  [ -f gpurun_out/bench_final_$t.log ] && tail -1 gpurun_out/bench_final_$t.log > profiles/${R}_${t}_bench.json
done
for t in c2 c3 c4; do
  fpl=256; [ $t = c4 ] && fpl=64
  python3 scripts/pmc_summarize.py final_$t $fpl > /dev/null && mv profiles/final_${t}_pmc.json profiles/${R}_${t}_pmc.json
done
ls -la profiles/
