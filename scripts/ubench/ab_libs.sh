# Same-box A/B of two builds of the library (same ABI): scripts/ubench/alt/prev.so against the tree's own, through
# tune_sweep.py (kernel times from the library's own stats).  usage (GPU box): bash scripts/ubench/ab_libs.sh <config> <res> <spp> [env...]
# order: current previous previous current
ROOT=${GRAFT_REPO_ROOT:-.}
cd $ROOT
cp rayrs_amd/librayrs_hip.so /tmp/cur.so
run() { cp $1 rayrs_amd/librayrs_hip.so; echo "== $2"; python scripts/ubench/tune_sweep.py $CFG $RES $SPP "" 2>&1 | grep -v "^compact" | tail -n ${LINES_OUT:-1}; }
CFG=${1:-5}; RES=${2:-2048}; SPP=${3:-1024}
run /tmp/cur.so current
run scripts/ubench/alt/prev.so previous
run scripts/ubench/alt/prev.so previous
run /tmp/cur.so current
cp /tmp/cur.so rayrs_amd/librayrs_hip.so
