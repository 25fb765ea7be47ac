#!/bin/bash
# usage (GPU box): bash scripts/fetch_calib.sh <tag>  -> gpurun_out/<tag>_fetch_calibration.json
# FETCH_SIZE / WRITE_SIZE (KiB) of scripts/ubench/fetch_calib.hip's three kernels against their known bytes.
TAG=${1:-calib}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out
mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $ROOT/scripts/ubench/fetch_calib $ROOT/scripts/ubench/fetch_calib.hip || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/calib_f -- $ROOT/scripts/ubench/fetch_calib > $OUT/calib_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/calib_w -- $ROOT/scripts/ubench/fetch_calib > $OUT/calib_w.log 2>&1
python - "$OUT" "$TAG" <<'PY'
import csv, glob, json, re, sys
out, tag = sys.argv[1], sys.argv[2]
known = {}
for line in open(out + "/calib_f.log"):
    m = re.match(r"gather(\d+): .* read (\d+) bytes .* written (\d+) bytes", line)
    if m: known[int(m.group(1))] = (int(m.group(2)), int(m.group(3)))
res = {}
for which, d in (("FETCH_SIZE", "calib_f"), ("WRITE_SIZE", "calib_w")):
    for f in glob.glob(f"{out}/{d}/*/*counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            m = re.search(r"gather<(\d+)>", row["Kernel_Name"])
            if not m or row["Counter_Name"] != which: continue
            rec = 16 * int(m.group(1))
            res.setdefault(rec, {})[which + "_bytes"] = float(row["Counter_Value"]) * 1024
for rec, r in res.items():
    r["read_bytes_known"], r["written_bytes_known"] = known[rec]
    r["fetch_over_known"] = round(r["FETCH_SIZE_bytes"] / known[rec][0], 4)
    r["write_over_known"] = round(r["WRITE_SIZE_bytes"] / known[rec][1], 4)
json.dump({"source": "scripts/fetch_calib.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) around "
           "scripts/ubench/fetch_calib.hip: random whole records from a 6 GiB table, 16-byte stores in order",
           "record_bytes": {str(k): v for k, v in sorted(res.items())}}, open(f"{out}/{tag}_fetch_calibration.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
rm -rf $OUT/calib_f $OUT/calib_w
