# Same-box A/B of two builds of the library on the headline frame: scripts/ubench/alt/prev.so against the tree's own.
# usage (GPU box): bash scripts/ubench/ab.sh   (order A B B A, kernel times from the library's own stats)
ROOT=${GRAFT_REPO_ROOT:-.}
cd $ROOT
cp rayrs_amd/librayrs_hip.so /tmp/cur.so
run() { cp $1 rayrs_amd/librayrs_hip.so; echo "== $2"; python scripts/perf_probe.py ${PROBE:-full5} ${PROBE_ARG:-} 2>&1 | tail -${LINES_OUT:-1}; }
run /tmp/cur.so current
run scripts/ubench/alt/prev.so previous
run scripts/ubench/alt/prev.so previous
run /tmp/cur.so current
cp /tmp/cur.so rayrs_amd/librayrs_hip.so
