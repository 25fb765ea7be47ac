# the bench lines again on the final tree (bench.py changed, the kernels did not: the PMC files still match)
set -e
mkdir -p gpurun_out/r5lines
python bench.py --no-build > gpurun_out/r5lines/final.json 2> gpurun_out/r5lines/final.err
for c in 1 2 3 4; do python bench.py --no-build --config $c > gpurun_out/r5lines/config$c.json 2> gpurun_out/r5lines/config$c.err; done
