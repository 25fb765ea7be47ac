set -e
mkdir -p gpurun_out/r5c
python scripts/ubench/cert_parity.py > gpurun_out/r5c/cert_parity.txt 2>&1
BUILD=0,1 python scripts/ubench/cert_parity.py > gpurun_out/r5c/cert_parity_whole.txt 2>&1
ONLY=certified,fast python scripts/ubench/exact_cost.py 5 2048 1024 > gpurun_out/r5c/walks.txt 2>&1
BUILD=0,1 ONLY=certified python scripts/ubench/exact_cost.py 5 2048 1024 > gpurun_out/r5c/walks_whole.txt 2>&1
BUILD=8,0 ONLY=certified python scripts/ubench/exact_cost.py 5 2048 1024 > gpurun_out/r5c/walks_w8.txt 2>&1
