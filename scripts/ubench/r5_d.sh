set -e
mkdir -p gpurun_out/r5d
ONLY=certified,fast python scripts/ubench/exact_cost.py 5 2048 1024 > gpurun_out/r5d/walks.txt 2>&1
BUILD=0,1 ONLY=certified python scripts/ubench/exact_cost.py 5 2048 1024 > gpurun_out/r5d/walks_whole.txt 2>&1
