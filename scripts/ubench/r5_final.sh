set -e
mkdir -p gpurun_out/r5z
python scripts/ubench/exact_cost.py 5 2048 1024 > gpurun_out/r5z/walks.txt 2>&1
CAMERA=close ONLY=default,fast python scripts/ubench/exact_cost.py 5 2048 1024 > gpurun_out/r5z/walks_close.txt 2>&1
python scripts/ubench/exact_cost.py 3 1024 512 > gpurun_out/r5z/walks_config3.txt 2>&1
python bench.py --no-build --share-of 8 --frames-in-flight 1 --steps 6 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/r5z/share8_one.json 2>gpurun_out/r5z/share8_one.err
python bench.py --no-build --share-of 8 --steps 6 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/r5z/share8_three.json 2>gpurun_out/r5z/share8_three.err
python scripts/ubench/trav_phases.py 5 2048 1024 > gpurun_out/r5z/trav_phases.txt 2>&1
