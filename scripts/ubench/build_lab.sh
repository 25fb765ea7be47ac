#!/bin/bash
# The LAB build of the library (make LAB=1: shader-clock shares of the hit / miss loops, rayrs_amd/csrc/lab_ticks.h), built outside
# the tree into scripts/ubench/alt_lab/lab.so; use it with RAYRS_HIP_LIB=.../lab.so python scripts/ubench/shade_ticks.py 5 2048 1024
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
B=/tmp/rayrs_lab_build
rm -rf $B && mkdir -p $B/include $B/rayrs_amd
cp $ROOT/include/*.h $B/include/
cp -r $ROOT/rayrs_amd/csrc $B/rayrs_amd/csrc
rm -f $B/rayrs_amd/csrc/*.o
make -C $B/rayrs_amd/csrc LAB=1 ../librayrs_hip.so > $B/build.log 2>&1 || { tail -20 $B/build.log; exit 1; }
mkdir -p $ROOT/scripts/ubench/alt_lab
cp $B/rayrs_amd/librayrs_hip.so $ROOT/scripts/ubench/alt_lab/lab.so
ls -la $ROOT/scripts/ubench/alt_lab/lab.so
