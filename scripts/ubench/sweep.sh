mkdir -p gpurun_out/r2
for T in "" "pool_slots=16777216" "pool_slots=50331648" "pool_slots=67108864" "refill_min=44" "refill_min=58" "leaf_min=16" "leaf_min=32" "leaf_min=40" "refill_min=44,leaf_min=32" "static_pct=25" "static_pct=75" "hot_records=64" "stack_lds=8"; do
  echo "== tuning: $T"
  PROBE_TUNING="$T" python scripts/perf_probe.py full5 2>&1 | tail -1
done
