for T in "pool_slots=1048576" "pool_slots=2097152" "pool_slots=4194304" "pool_slots=8388608"; do
  echo "== tuning: $T"
  PROBE_TUNING="$T" python scripts/perf_probe.py full5 2>&1 | tail -1
done
