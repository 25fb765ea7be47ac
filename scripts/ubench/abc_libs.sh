# Same-box comparison of several builds of the library (same ABI): every scripts/ubench/alt/*.so and the tree's own,
# through tune_sweep.py, in the order A B C ... C B A.  usage (GPU box): bash scripts/ubench/abc_libs.sh <config> <res> <spp>
# The build is chosen with RAYRS_HIP_LIB (rayrs_amd/_ffi.py); nothing in the tree is overwritten.
ROOT=${GRAFT_REPO_ROOT:-.}
cd $ROOT
CFG=${1:-5}; RES=${2:-2048}; SPP=${3:-1024}
LIBS="$PWD/rayrs_amd/librayrs_hip.so $(ls $PWD/scripts/ubench/alt/*.so)"
REV=$(echo $LIBS | tr ' ' '\n' | tac | tr '\n' ' ')
for l in $LIBS $REV; do
  echo "== $l"
  RAYRS_HIP_LIB=$l python scripts/ubench/tune_sweep.py $CFG $RES $SPP "${SET:-}" 2>&1 | grep -v "^compact" | tail -n 1
done
