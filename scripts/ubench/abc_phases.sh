# trav_phases.py for the tree's library and every scripts/ubench/alt/*.so (GPU box)
ROOT=${GRAFT_REPO_ROOT:-.}
cd $ROOT
CFG=${1:-5}; RES=${2:-2048}; SPP=${3:-1024}
cp rayrs_amd/librayrs_hip.so /tmp/cur.so
for l in /tmp/cur.so $(ls scripts/ubench/alt/*.so); do
  cp $l rayrs_amd/librayrs_hip.so; echo "== $l"
  python scripts/ubench/trav_phases.py $CFG $RES $SPP 2>&1 | tail -n 5
done
cp /tmp/cur.so rayrs_amd/librayrs_hip.so
