#!/bin/bash
# usage (GPU box): scripts/pmc.sh <tag> <probe-arg | bench[:extra bench.py flags]> "<counters pass1>" "<counters pass2>" ...
# Prints, per kernel, the sum of every counter over its dispatches.  Each pass is its own rocprofv3 run
# with --kernel-trace only (never combined with other trace domains).  The library must be built already:
# a profiled process must not start the compiler.
TAG=$1; ARG=$2; shift 2
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
python -c 'import __graft_entry__ as g; g.build()' || exit 1
cd /tmp && export TMPDIR=/tmp
case "$ARG" in
  bench*) X=${ARG#bench}; X=${X#:}; CMD="python $ROOT/bench.py --no-build --steps 1 --warmup 0 --no-cpu-baseline --no-roofline $X";;
  *)      CMD="python $ROOT/scripts/perf_probe.py $ARG";;
esac
i=0
for P in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/p$i -- $CMD > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/p$i.log; }
done
python - "$OUT" <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for f in glob.glob(out + "/p*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'].split('(')[0].replace('void ', '').replace('rayrs::', '')[:28]
        agg[k][row['Counter_Name']] += float(row['Counter_Value'])
        disp[k].add(row['Dispatch_Id'])
dur = collections.defaultdict(int)
for f in glob.glob(out + "/p1/*/*kernel_trace.csv"):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'].split('(')[0].replace('void ', '').replace('rayrs::', '')[:28]
        dur[k] += int(row['End_Timestamp']) - int(row['Start_Timestamp'])
for k in sorted(agg, key=lambda k: -dur[k]):
    if not k.startswith(('wf_', 'lp_')): continue
    print(f"== {k}: {len(disp[k])} dispatches, {dur[k]/1e6:.1f} ms (pass 1) :: " + " ".join(f"{c}={agg[k][c]:.5g}" for c in sorted(agg[k])))
PY
find $OUT -name "*.csv" -size +20M -delete
