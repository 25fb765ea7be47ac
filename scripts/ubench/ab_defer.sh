# Same-box A/B of the tree's library against scripts/ubench/alt/prev.so (build_prev.sh) with a sweep of the default walk's leaf-phase
# thresholds on the tree's library.  usage (GPU box): bash scripts/ubench/ab_defer.sh <config> <res> <spp> "k=v,.." ...
ROOT=${GRAFT_REPO_ROOT:-.}
cd $ROOT
CFG=${1:-5}; RES=${2:-2048}; SPP=${3:-1024}; shift 3
CUR=$PWD/rayrs_amd/librayrs_hip.so; PREV=$PWD/scripts/ubench/alt/prev.so
echo "== previous"; RAYRS_HIP_LIB=$PREV python scripts/ubench/tune_sweep.py $CFG $RES $SPP "" 2>&1 | grep -v "^compact"
echo "== current"; RAYRS_HIP_LIB=$CUR python scripts/ubench/tune_sweep.py $CFG $RES $SPP "$@" 2>&1 | grep -v "^compact"
echo "== previous"; RAYRS_HIP_LIB=$PREV python scripts/ubench/tune_sweep.py $CFG $RES $SPP "" 2>&1 | grep -v "^compact"
