for T in "" "pool_slots=16777216" "pool_slots=8388608" "pool_slots=25165824"; do
  echo "== tuning: $T"
  PROBE_TUNING="$T" python scripts/perf_probe.py shard8 4 2>&1 | tail -1
done
