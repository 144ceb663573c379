# After `gpurun -- bash scripts/gpu_round_end.sh <round>`: copy the judged summaries from gpurun_out/ (scratch) into
# profiles/ (tracked).  usage: bash scripts/collect_profiles.sh [round_tag, default round5]
set -e
R=${1:-round6}
cd "$(dirname "$0")/.."
for t in c1 c1ref c2 c3 c2ff c4 c3fb c3noise c3ref c3full; do
  f=$(ls -t gpurun_out/prof_final_$t/*/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && cp "$f" profiles/${R}_${t}_kernel_stats.csv
  [ -f gpurun_out/bench_final_$t.log ] && tail -1 gpurun_out/bench_final_$t.log > profiles/${R}_${t}_bench.json
  [ -f gpurun_out/bench_final_${t}_serial.log ] && grep '^{' gpurun_out/bench_final_${t}_serial.log | tail -1 > profiles/${R}_${t}_serial_bench.json
done
for t in c2 c3 c4; do
  [ -f gpurun_out/${R}_${t}_pmc.json ] && cp gpurun_out/${R}_${t}_pmc.json profiles/${R}_${t}_pmc.json
done
[ -f gpurun_out/${R}_c3_valu.json ] && cp gpurun_out/${R}_c3_valu.json profiles/${R}_c3_valu.json
for res in pinned pageable resident; do   # scripts/gpu_api_trace.sh
  [ -f gpurun_out/${R}_api_trace_$res.json ] && cp gpurun_out/${R}_api_trace_$res.json profiles/${R}_api_trace_$res.json
  [ -f gpurun_out/${R}_api_trace_${res}_kernel_stats.csv ] && cp gpurun_out/${R}_api_trace_${res}_kernel_stats.csv profiles/${R}_api_trace_${res}_kernel_stats.csv
done
[ -f gpurun_out/${R}_api_untraced.log ] && grep -v amdgpu gpurun_out/${R}_api_untraced.log > profiles/${R}_api_untraced.log
[ -f gpurun_out/clock_trace.json ] && [ gpurun_out/clock_trace.json -nt profiles/README.md ] && cp gpurun_out/clock_trace.json profiles/${R}_c3_clock.json
# (older clock traces were copied by hand: gpurun_out/ accumulates across rounds, and scripts/clock_trace.py / ssim_clock.py are
#  separate runs: cp gpurun_out/clock_trace.json profiles/${R}_c3_clock.json after running them in THIS round)
ls -la profiles/ | grep ${R}
