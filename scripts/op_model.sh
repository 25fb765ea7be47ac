#!/bin/bash
# usage (GPU box): bash scripts/op_model.sh <tag>   ->  gpurun_out/<tag>_op_model.json
# Vector instructions per evaluation of every shading unit compiled alone at 64/64 lanes (scripts/op_model.py),
# from one rocprofv3 --pmc SQ_INSTS_VALU pass (with --kernel-trace only).  The library must be built already.
TAG=${1:-r03}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/opm_$TAG
mkdir -p $OUT
python -c 'import __graft_entry__ as g; g.check_built()' || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/pmc -- python $ROOT/scripts/op_model.py $OUT/order.json > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
python - "$OUT" "$ROOT/gpurun_out/${TAG}_op_model.json" "$ROOT" <<'PY'
import csv, glob, json, sys
out, dst = sys.argv[1], sys.argv[2]
sys.path.insert(0, sys.argv[3])
import bench
order = json.load(open(out + "/order.json"))
rows = []
for f in glob.glob(out + "/pmc/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] == "SQ_INSTS_VALU" and row["Kernel_Name"].split("(")[0].split("::")[-1].startswith(("test_material_kernel", "test_background_kernel")):
            rows.append((int(row["Dispatch_Id"]), row["Kernel_Name"], float(row["Counter_Value"])))
rows.sort()
assert len(rows) == len(order), (len(rows), len(order))
res = {}
for (did, kname, val), o in zip(rows, order):
    assert o["kernel"] in kname, (kname, o)
    res[o["unit"]] = {"valu_wave_instructions": val, "evaluations": o["n"], "valu_per_evaluation": round(val * 64 / o["n"], 1),
                      **{k: o[k] for k in ("scattered", "draws") if k in o}}
base = res["none"]["valu_per_evaluation"]
for k, v in res.items():
    if k not in ("none", "background"):
        v["valu_per_evaluation_net"] = round(v["valu_per_evaluation"] - base, 1)
json.dump({"shading_source_hash": bench.shading_source_hash(), "source_hash": bench.source_hash(),
           "source": "scripts/op_model.sh: rocprofv3 --pmc SQ_INSTS_VALU --kernel-trace around scripts/op_model.py; "
                     "valu_per_evaluation = SQ_INSTS_VALU x 64 / evaluations (all 64 lanes of every wave active); "
                     "_net = minus the `none` unit (the test kernel's own loads and stores)", "units": res}, open(dst, "w"), indent=1)
for k, v in res.items():
    print(f"{k:28s} {v['valu_per_evaluation']:8.1f}  net {v.get('valu_per_evaluation_net', '')}")
PY
rm -rf $OUT/pmc
