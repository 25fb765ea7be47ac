set -e
mkdir -p gpurun_out/r5t
python scripts/ubench/tune_sweep.py 5 2048 1024 "" "leaf_min=28" "leaf_min=36" "refill_min=48" "refill_min=56" "hot_records=128" "static_pct=25" "trav_blocks_per_cu=4" "pool_slots=134217728" > gpurun_out/r5t/default_walk_sweep.txt 2>&1
