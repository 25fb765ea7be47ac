#!/bin/bash
# usage (on the GPU box, through gpurun): bash scripts/profile_round.sh <tag>
# Produces, under gpurun_out/<tag>/: bench.json (the full default bench line), kernel_stats.csv
# (rocprofv3 --kernel-trace --stats of one bench step) and hbm_traffic.json (FETCH_SIZE and
# WRITE_SIZE, separate --pmc passes, summed per kernel).  Copy what should be judged to profiles/.
set -e
TAG=${1:-round}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 600 $OUT/bench.json; echo
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/stats.log 2>&1
cp $OUT/stats/*/*kernel_stats.csv $OUT/kernel_stats.csv
echo "stats done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- $CMD > $OUT/fetch.log 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- $CMD > $OUT/write.log 2>&1
echo "write done"
python - "$OUT" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for which in ("fetch", "write"):
    for f in glob.glob(f"{out}/{which}/*/*counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("rayrs::", "").split("<")[0]
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
            disp[k].add((which, row["Dispatch_Id"]))
res = {}
for k in agg:
    n = len({d for w, d in disp[k] if w == "fetch"})
    res[k] = {"launches": n, "fetch_KiB": agg[k].get("FETCH_SIZE", 0.0), "write_KiB": agg[k].get("WRITE_SIZE", 0.0),
              "bytes_per_launch": int((agg[k].get("FETCH_SIZE", 0.0) + agg[k].get("WRITE_SIZE", 0.0)) * 1024 / max(n, 1))}
bench = json.loads(open(out + "/bench.json").read().strip().splitlines()[-1])
json.dump({"workload": bench["config"]["workload"], "sample_chunk": bench["config"]["sample_chunk"], "kernels": res},
          open(out + "/hbm_traffic.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
rm -rf $OUT/fetch $OUT/write $OUT/stats
