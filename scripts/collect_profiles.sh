# After `gpurun -- bash scripts/gpu_round_end.sh`: copy the judged summaries from gpurun_out/ (scratch) into
# profiles/ (tracked).  usage: bash scripts/collect_profiles.sh [round_tag, default round2]
set -e
R=${1:-round2}
cd "$(dirname "$0")/.."
for t in c2 c3 c2ff c4 c3fb c3noise; do
  f=$(ls -t gpurun_out/prof_final_$t/*/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && cp "$f" profiles/${R}_${t}_kernel_stats.csv
  [ -f gpurun_out/bench_final_$t.log ] && tail -1 gpurun_out/bench_final_$t.log > profiles/${R}_${t}_bench.json
done
for t in c2 c3 c4; do
  [ -f gpurun_out/final_${t}_pmc.json ] && sed "s/\"tag\": \"final_/\"tag\": \"${R}_/" gpurun_out/final_${t}_pmc.json > profiles/${R}_${t}_pmc.json
done
[ -f gpurun_out/final_c3_valu.json ] && sed "s/\"tag\": \"final_/\"tag\": \"${R}_/" gpurun_out/final_c3_valu.json > profiles/${R}_c3_valu.json
rm -f profiles/final_*
ls -la profiles/
