set -e
mkdir -p gpurun_out/r5a
python scripts/ubench/exact_cost.py 5 2048 1024 > gpurun_out/r5a/exact_cost.txt 2>&1
EXACT=1 python scripts/ubench/tune_sweep.py 5 2048 1024 "" "leaf_min=16" "leaf_min=32" "leaf_min=40" "leaf_min=48" "refill_min=44" "refill_min=58" "hot_records=64" "stack_lds=8" "stack_lds=16" > gpurun_out/r5a/exact_sweep.txt 2>&1
python scripts/ubench/trav_phases.py 5 2048 1024 > gpurun_out/r5a/trav_phases_default.txt 2>&1
