#!/bin/bash
# Builds the library of another commit (default HEAD) outside the tree into scripts/ubench/alt/prev.so, for the same-box A/B of
# scripts/ubench/ab_libs.sh / abc_libs.sh (the build is chosen there with RAYRS_HIP_LIB).  usage: bash scripts/ubench/build_prev.sh [commit]
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
REV=${1:-HEAD}
B=/tmp/rayrs_prev_build
rm -rf $B && mkdir -p $B
git -C $ROOT archive $REV include rayrs_amd/csrc | tar -x -C $B
make -C $B/rayrs_amd/csrc ../librayrs_hip.so > $B/build.log 2>&1 || { tail -20 $B/build.log; exit 1; }
mkdir -p $ROOT/scripts/ubench/alt
cp $B/rayrs_amd/librayrs_hip.so $ROOT/scripts/ubench/alt/prev.so
echo "$(git -C $ROOT rev-parse --short $REV) -> scripts/ubench/alt/prev.so"
