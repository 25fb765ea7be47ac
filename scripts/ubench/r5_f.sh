set -e
mkdir -p gpurun_out/r5f
python scripts/ubench/cert_parity.py > gpurun_out/r5f/cert_parity.txt 2>&1
ONLY=certified,reference python scripts/ubench/exact_cost.py 5 2048 1024 leaf_min=32 > gpurun_out/r5f/walks_leaf32.txt 2>&1
