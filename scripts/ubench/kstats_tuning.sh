#!/bin/bash
# usage (GPU box): bash scripts/ubench/kstats_tuning.sh <tag> <config> <res> <spp> "<tuning k=v,...>"
# rocprofv3 --kernel-trace --stats of one frame rendered with the given rayrs_tuning (library built beforehand).
TAG=$1; CFG=$2; RES=$3; SPP=$4; TUNE=$5
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks_$TAG -- python $ROOT/scripts/ubench/tune_sweep.py $CFG $RES $SPP "$TUNE" > $OUT/ks_$TAG.log 2>&1
cp $OUT/ks_$TAG/*/*kernel_stats.csv $OUT/${TAG}_kernel_stats.csv
rm -rf $OUT/ks_$TAG
echo "== $TAG: $TUNE"
cut -d, -f1-4 $OUT/${TAG}_kernel_stats.csv | sed 's/rayrs:://; s/(.*)//' | cut -c1-120 | head -8
