mkdir -p gpurun_out/r2
for T in "" "pipelines=2" "pipelines=2,trav_blocks_per_cu=4" "pipelines=2,trav_blocks_per_cu=3" "trav_blocks_per_cu=4"; do
  echo "== tuning: $T"
  PROBE_TUNING="$T" python scripts/perf_probe.py full5 2>&1 | tail -1
done
