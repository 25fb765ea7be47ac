set -e
mkdir -p gpurun_out/r5e
python scripts/ubench/cert_parity.py > gpurun_out/r5e/cert_parity.txt 2>&1
ONLY=certified,reference,fast python scripts/ubench/exact_cost.py 5 2048 1024 > gpurun_out/r5e/walks.txt 2>&1
ONLY=certified,reference python scripts/ubench/exact_cost.py 5 2048 1024 leaf_min=32 > gpurun_out/r5e/walks_leaf32.txt 2>&1
