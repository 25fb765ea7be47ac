# rocprofv3 kernel stats of one tune_sweep run of the tree's library (GPU box): bash scripts/ubench/kstats_lib.sh <config> <res> <spp> ["k=v,..."]
ROOT=${GRAFT_REPO_ROOT:-.}
CFG=${1:-5}; RES=${2:-2048}; SPP=${3:-1024}; SET=${4:-}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kst
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kst -- python $ROOT/scripts/ubench/tune_sweep.py $CFG $RES $SPP "$SET" > /tmp/kst.log 2>&1
tail -n 2 /tmp/kst.log
python - <<'PY'
import csv, glob
for f in glob.glob("/tmp/kst/*/*kernel_stats.csv"):
    for row in csv.DictReader(open(f)):
        n = row["Name"].split("(")[0].replace("void ", "").replace("rayrs::", "")
        print(f"{n:60s} calls {row['Calls']:>6s} total {float(row['TotalDurationNs'])/1e6:10.2f} ms avg {float(row['AverageNs'])/1e3:10.1f} us  {row['Percentage']}%")
PY
