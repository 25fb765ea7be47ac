# Same-box A/B of two builds of the library (same ABI): scripts/ubench/alt/prev.so against the tree's own.
# usage (GPU box): bash scripts/ubench/ab.sh ; order: current previous previous current.
# The build is chosen with RAYRS_HIP_LIB (rayrs_amd/_ffi.py); nothing in the tree is overwritten.
ROOT=${GRAFT_REPO_ROOT:-.}
cd $ROOT
run() { echo "== $2"; RAYRS_HIP_LIB=$1 python scripts/perf_probe.py ${PROBE:-full5} ${PROBE_ARG:-} 2>&1 | tail -${LINES_OUT:-1}; }
CUR=$PWD/rayrs_amd/librayrs_hip.so; PREV=$PWD/scripts/ubench/alt/prev.so
run $CUR current
run $PREV previous
run $PREV previous
run $CUR current
