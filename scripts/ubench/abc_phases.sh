# trav_phases.py for the tree's library and every scripts/ubench/alt/*.so (GPU box); the build is chosen with RAYRS_HIP_LIB
ROOT=${GRAFT_REPO_ROOT:-.}
cd $ROOT
CFG=${1:-5}; RES=${2:-2048}; SPP=${3:-1024}
for l in $PWD/rayrs_amd/librayrs_hip.so $(ls $PWD/scripts/ubench/alt/*.so); do
  echo "== $l"
  RAYRS_HIP_LIB=$l python scripts/ubench/trav_phases.py $CFG $RES $SPP 2>&1 | tail -n 6
done
