set -e
mkdir -p gpurun_out/r5b
python scripts/ubench/cert_parity.py > gpurun_out/r5b/cert_parity.txt 2>&1
python scripts/ubench/exact_cost.py 5 2048 1024 > gpurun_out/r5b/walks.txt 2>&1
