# Same-box comparison of several builds of the library (same ABI): every scripts/ubench/alt/*.so and the tree's own,
# through tune_sweep.py, in the order A B C ... C B A.  usage (GPU box): bash scripts/ubench/abc_libs.sh <config> <res> <spp>
ROOT=${GRAFT_REPO_ROOT:-.}
cd $ROOT
CFG=${1:-5}; RES=${2:-2048}; SPP=${3:-1024}
cp rayrs_amd/librayrs_hip.so /tmp/cur.so
LIBS="/tmp/cur.so $(ls scripts/ubench/alt/*.so)"
REV=$(echo $LIBS | tr ' ' '\n' | tac | tr '\n' ' ')
for l in $LIBS $REV; do
  cp $l rayrs_amd/librayrs_hip.so; echo "== $l"
  python scripts/ubench/tune_sweep.py $CFG $RES $SPP "" 2>&1 | grep -v "^compact" | tail -n 1
done
cp /tmp/cur.so rayrs_amd/librayrs_hip.so
