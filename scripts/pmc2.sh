#!/bin/bash
# usage: scripts/pmc2.sh <tag> <probe-arg> <kernel-name-pattern> "<counters pass1>" "<counters pass2>" ...
TAG=$1; ARG=$2; PAT=$3; shift 3
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for P in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/p$i -- python $GRAFT_REPO_ROOT/scripts/perf_probe.py $ARG > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/p$i.log; }
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob("$OUT/p*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        if '' in row['Kernel_Name']:
            agg[row['Counter_Name']] += float(row['Counter_Value']); n[row['Counter_Name']] += 1
for k in sorted(agg): print(f"{k:32s} {agg[k]:.6g}  (dispatches {n[k]})")
tot=0
for f in glob.glob("$OUT/p1/*/*kernel_trace.csv"):
    for row in csv.DictReader(open(f)):
        if '' in row['Kernel_Name']:
            tot += int(row['End_Timestamp'])-int(row['Start_Timestamp'])
print("trace_kernel total ns (pass 1):", tot)
PY
