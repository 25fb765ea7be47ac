# kernel timeline of the two-pipeline mode: do the shading kernels of one half really run beside the
# other half's traversal?  usage: bash scripts/ubench/pipes_trace.sh "<tuning>"
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/pipes_trace
mkdir -p $OUT
python -c 'import __graft_entry__ as g; g.build()' || exit 1
cd /tmp && export TMPDIR=/tmp
export PROBE_TUNING="$1"
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python $ROOT/scripts/perf_probe.py full5 > $OUT/run.log 2>&1
tail -1 $OUT/run.log
python - $OUT <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/t/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        k = 'trav' if 'wf_trav' in n else 'hit' if 'wf_hit' in n else 'miss' if 'wf_miss' in n else None
        if k: rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), k, r.get('Queue_Id', '?')))
rows.sort()
# keep the long frame only (last 60 % of the dispatches)
rows = rows[len(rows) // 3:]
t0 = rows[0][0]
tot = {}
for s, e, k, q in rows: tot[k] = tot.get(k, 0) + (e - s)
span = rows[-1][1] - t0
# time during which a trav kernel and a shading kernel are both in flight
ev = []
for s, e, k, q in rows:
    ev.append((s, 1, k)); ev.append((e, -1, k))
ev.sort()
cnt = {'trav': 0, 'hit': 0, 'miss': 0}; last = ev[0][0]; both = 0; anytrav = 0; anyshade = 0
for t, d, k in ev:
    dt = t - last; last = t
    if cnt['trav'] > 0: anytrav += dt
    if cnt['hit'] + cnt['miss'] > 0: anyshade += dt
    if cnt['trav'] > 0 and cnt['hit'] + cnt['miss'] > 0: both += dt
    cnt[k] += d
print(f"span {span/1e6:.1f} ms; kernel time sums: " + " ".join(f"{k} {v/1e6:.1f}" for k, v in tot.items()))
print(f"trav in flight {anytrav/1e6:.1f} ms, shading in flight {anyshade/1e6:.1f} ms, both {both/1e6:.1f} ms")
mid = len(rows) // 2
for s, e, k, q in rows[mid:mid + 14]:
    print(f"  {k:5s} q{q} start {(s-t0)/1e6:9.3f} dur {(e-s)/1e6:7.3f}")
PY
rm -rf $OUT/t
