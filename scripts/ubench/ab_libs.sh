# Same-box A/B of several builds of the library: usage (GPU box): bash scripts/ubench/ab_libs.sh "<tune_sweep args>" lib1.so lib2.so ...
# Each library (paths relative to the repo root; "tree" = the tree's own) renders the sweep once, in the order
# given and then reversed.
ROOT=${GRAFT_REPO_ROOT:-.}
cd $ROOT
ARGS=$1; shift
cp rayrs_amd/librayrs_hip.so /tmp/tree.so
run() { if [ "$1" = tree ]; then cp /tmp/tree.so rayrs_amd/librayrs_hip.so; else cp $1 rayrs_amd/librayrs_hip.so; fi; echo "== $1"; python scripts/ubench/tune_sweep.py $ARGS 2>&1 | grep trace | head -1; }
LIBS=("$@")
for l in "${LIBS[@]}"; do run $l; done
for ((i=${#LIBS[@]}-1; i>=0; i--)); do run ${LIBS[$i]}; done
cp /tmp/tree.so rayrs_amd/librayrs_hip.so
