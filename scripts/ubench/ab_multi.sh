# Same-box comparison of several builds of the library, each with its own list of rayrs_lab.h settings (tune_sweep.py: A B .. B A).
# usage (GPU box): bash scripts/ubench/ab_multi.sh <config> <res> <spp> lib1.so "set1|set2|.." lib2.so "set1|.." ...   ("-" = the tree's library)
ROOT=${GRAFT_REPO_ROOT:-.}
cd $ROOT
CFG=$1; RES=$2; SPP=$3; shift 3
while [ $# -ge 2 ]; do
  LIB=$1; SETS=$2; shift 2
  [ "$LIB" = "-" ] && LIB=$PWD/rayrs_amd/librayrs_hip.so
  IFS='|' read -ra ARR <<< "$SETS"
  echo "== $LIB"
  RAYRS_HIP_LIB=$LIB python scripts/ubench/tune_sweep.py $CFG $RES $SPP "${ARR[@]}" 2>&1 | grep -v "^compact"
done
