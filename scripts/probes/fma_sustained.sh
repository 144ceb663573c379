# sustained fp32 vector FMA rate + the clock / power rocm-smi shows meanwhile (GPU box): bash scripts/probes/fma_sustained.sh
for mode in 0 1; do
  ./gpurun_ab/fma_sustained 6 $mode &
  pid=$!
  sleep 3
  /opt/rocm/bin/rocm-smi -d 0 --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | head -3
  wait $pid
done
