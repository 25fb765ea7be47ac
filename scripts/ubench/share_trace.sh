# per-round kernel times of a one-eighth tile share of the headline frame (where do the 26 ms over 1/8 of a frame go?)
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/share_trace
mkdir -p $OUT
python -c 'import __graft_entry__ as g; g.build()' || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python $ROOT/scripts/perf_probe.py ${PROBE:-shard8} ${PROBE_ARG-4} > $OUT/run.log 2>&1
tail -1 $OUT/run.log
python - $OUT <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/t/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        k = 'trav' if 'wf_trav' in n else 'hit' if 'wf_hit' in n else 'miss' if 'wf_miss' in n else 'gen' if 'wf_gen' in n else None
        if k: rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), k))
rows.sort()
# the last render = last gen kernel onwards
g = max(i for i, r in enumerate(rows) if r[2] == 'gen')
rows = rows[g:]
t0 = rows[0][0]
rnd = []
cur = {}
for s, e, k in rows[1:]:
    if k == 'trav' and cur: rnd.append(cur); cur = {}
    cur[k] = (e - s) / 1e6; cur.setdefault('start', (s - t0) / 1e6); cur['end'] = (e - t0) / 1e6
rnd.append(cur)
print("gen %.2f ms" % ((rows[0][1] - rows[0][0]) / 1e6))
for i, r in enumerate(rnd):
    print(f"round {i:3d} start {r['start']:7.2f}  trav {r.get('trav',0):6.2f} hit {r.get('hit',0):6.2f} miss {r.get('miss',0):6.2f}  gap-free sum {r.get('trav',0)+r.get('hit',0)+r.get('miss',0):6.2f}  span {r['end']-r['start']:6.2f}")
PY
rm -rf $OUT/t
