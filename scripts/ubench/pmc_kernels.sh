# Issue-side PMC (two passes; a third with seven TCC counters did not come back within seven minutes) per kernel for one tune_sweep run of the tree's library (GPU box).
# usage: bash scripts/ubench/pmc_kernels.sh <config> <res> <spp> ["k=v,..."]
ROOT=${GRAFT_REPO_ROOT:-.}
CFG=${1:-5}; RES=${2:-2048}; SPP=${3:-1024}; SET=${4:-}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_a /tmp/pmc_b /tmp/pmc_c
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU --kernel-trace --output-format csv -d /tmp/pmc_a -- python $ROOT/scripts/ubench/tune_sweep.py $CFG $RES $SPP "$SET" > /tmp/pmc_a.log 2>&1
echo "issue pass done"
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_FLAT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d /tmp/pmc_b -- python $ROOT/scripts/ubench/tune_sweep.py $CFG $RES $SPP "$SET" > /tmp/pmc_b.log 2>&1
tail -n 1 /tmp/pmc_a.log
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for d in ("/tmp/pmc_a", "/tmp/pmc_b", "/tmp/pmc_c"):
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("rayrs::", "")
            if not k.startswith("wf_"): continue
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); disp[k].add((d, row["Dispatch_Id"]))
for k, c in sorted(agg.items()):
    n = max(len({x for dd, x in disp[k] if dd == "/tmp/pmc_a"}), 1)
    print(k, "launches", n)
    print("   " + " ".join(f"{a}={v / n:.4g}" for a, v in sorted(c.items())))
    if c["GRBM_GUI_ACTIVE"]:
        print("   valu_busy", round(c["SQ_INSTS_VALU"] * 4 / (c["GRBM_GUI_ACTIVE"] / 8 * 1024), 4),
              "wait_any/wave_cycles", round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 4),
              "waves resident per SIMD", round(c["SQ_WAVE_CYCLES"] * 4 / (c["GRBM_GUI_ACTIVE"] / 8 * 1024), 2))
PY
