#!/bin/bash
# usage (GPU box): bash scripts/valu_pass.sh <tag>   -> gpurun_out/<tag>/valu.json
# FP64-VALU occupancy of each kernel of one bench step: wave-instructions issued (SQ_INSTS_VALU, 4 cycles
# each on a 16-lane SIMD) against the SIMD-cycles the kernel had (GRBM_GUI_ACTIVE is summed over the 8 XCDs,
# 128 SIMDs each).
set -e
TAG=${1:-round}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline"
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/valu -- $CMD > $OUT/valu.log 2>&1
python - "$OUT" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for f in glob.glob(f"{out}/valu/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("rayrs::", "").split("<")[0]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); disp[k].add(row["Dispatch_Id"])
res = {}
for k, c in agg.items():
    if not k.startswith("wf_"): continue
    simd_cycles = c["GRBM_GUI_ACTIVE"] / 8 * 1024
    res[k] = {"launches": len(disp[k]), "valu_wave_instructions": c["SQ_INSTS_VALU"], "salu_wave_instructions": c["SQ_INSTS_SALU"],
              "gpu_cycles_per_xcd": c["GRBM_GUI_ACTIVE"] / 8, "valu_busy": round(c["SQ_INSTS_VALU"] * 4 / simd_cycles, 4),
              "valu_busy_active_inst": round(c["SQ_ACTIVE_INST_VALU"] * 4 / simd_cycles, 4)}
json.dump({"kernels": res}, open(out + "/valu.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
rm -rf $OUT/valu
