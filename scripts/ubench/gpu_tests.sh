# the GPU test suite only (development aid: gpurun -- 'bash scripts/ubench/gpu_tests.sh TAG')
TAG=${1:-tests}
mkdir -p gpurun_out/$TAG
python -m pytest tests -m gpu -x -q > gpurun_out/$TAG/pytest_gpu.txt 2>&1
rc=$?
tail -40 gpurun_out/$TAG/pytest_gpu.txt
exit $rc
