# the GPU test suite, then the default bench line (development aid: gpurun -- 'bash scripts/ubench/gpu_check.sh TAG')
set -e
TAG=${1:-check}
mkdir -p gpurun_out/$TAG
python -m pytest tests -m gpu -x -q > gpurun_out/$TAG/pytest_gpu.txt 2>&1 || { tail -30 gpurun_out/$TAG/pytest_gpu.txt; exit 1; }
tail -3 gpurun_out/$TAG/pytest_gpu.txt
python bench.py --no-build > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err || { tail -20 gpurun_out/$TAG/bench.err; exit 1; }
