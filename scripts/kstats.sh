#!/bin/bash
# usage (GPU box): bash scripts/kstats.sh <tag> [bench.py flags]   -> gpurun_out/<tag>_kernel_stats.csv
# rocprofv3 --kernel-trace --stats of one bench step (library built beforehand, outside the profiler).
TAG=$1; shift
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out
mkdir -p $OUT
python -c 'import __graft_entry__ as g; g.build()' || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks_$TAG -- python $ROOT/bench.py --no-build --steps 1 --warmup 0 --no-cpu-baseline --no-roofline "$@" > $OUT/ks_$TAG.log 2>&1
cp $OUT/ks_$TAG/*/*kernel_stats.csv $OUT/${TAG}_kernel_stats.csv
cp $OUT/ks_$TAG/*/*kernel_trace.csv $OUT/${TAG}_kernel_trace.csv
rm -rf $OUT/ks_$TAG
cut -d, -f1-4 $OUT/${TAG}_kernel_stats.csv | cut -c1-200 | head -8
tail -c 400 $OUT/ks_$TAG.log | grep -o '"value": [0-9.]*' 
