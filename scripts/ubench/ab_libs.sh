# Same-box A/B of two builds of the library (same ABI): scripts/ubench/alt/prev.so against the tree's own, through
# tune_sweep.py (kernel times from the library's own stats).  usage (GPU box): bash scripts/ubench/ab_libs.sh <config> <res> <spp> [env...]
# order: current previous previous current.  The build is chosen with RAYRS_HIP_LIB (rayrs_amd/_ffi.py); nothing in the tree is overwritten.
ROOT=${GRAFT_REPO_ROOT:-.}
cd $ROOT
run() { echo "== $2"; RAYRS_HIP_LIB=$1 python scripts/ubench/tune_sweep.py $CFG $RES $SPP "" 2>&1 | grep -v "^compact" | tail -n ${LINES_OUT:-1}; }
CFG=${1:-5}; RES=${2:-2048}; SPP=${3:-1024}
CUR=$PWD/rayrs_amd/librayrs_hip.so; PREV=$PWD/scripts/ubench/alt/prev.so
run $CUR current
run $PREV previous
run $PREV previous
run $CUR current
