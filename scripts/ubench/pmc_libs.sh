# Issue-side PMC of the traversal kernel for the tree's library and every scripts/ubench/alt/*.so on one frame (GPU box).
# usage: bash scripts/ubench/pmc_libs.sh <config> <res> <spp>
ROOT=${GRAFT_REPO_ROOT:-.}
CFG=${1:-5}; RES=${2:-2048}; SPP=${3:-1024}
# (the build is chosen with RAYRS_HIP_LIB, exported before rocprofv3 -- no `env` hop behind the profiler; nothing in the tree is overwritten)
ROOT=$(cd $ROOT && pwd)
cd /tmp && export TMPDIR=/tmp
for l in $ROOT/rayrs_amd/librayrs_hip.so $(ls $ROOT/scripts/ubench/alt/*.so); do
  export RAYRS_HIP_LIB=$l; echo "== $l"
  rm -rf /tmp/pmc_sq
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU --kernel-trace --output-format csv -d /tmp/pmc_sq -- python $ROOT/scripts/ubench/tune_sweep.py $CFG $RES $SPP "" > /tmp/pmc_sq.log 2>&1
  tail -n 1 /tmp/pmc_sq.log
  rm -rf /tmp/pmc_sq2
  rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_FLAT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d /tmp/pmc_sq2 -- python $ROOT/scripts/ubench/tune_sweep.py $CFG $RES $SPP "" > /tmp/pmc_sq2.log 2>&1
  python - <<'PY'
import csv, glob, collections
for d in ("/tmp/pmc_sq", "/tmp/pmc_sq2"):
    agg = collections.defaultdict(float); disp = set()
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            if "wf_trav_kernel" not in row["Kernel_Name"]: continue
            agg[row["Counter_Name"]] += float(row["Counter_Value"]); disp.add(row["Dispatch_Id"])
    n = max(len(disp), 1)
    print("  launches", n, " ".join(f"{k}={v / n:.4g}" for k, v in sorted(agg.items())))
    if "GRBM_GUI_ACTIVE" in agg:
        print("  valu_busy", round(agg["SQ_INSTS_VALU"] * 4 / (agg["GRBM_GUI_ACTIVE"] / 8 * 1024), 4),
              "wait_any/wave_cycles", round(agg["SQ_WAIT_ANY"] / agg["SQ_WAVE_CYCLES"], 4),
              "wait_inst_any/wave_cycles", round(agg["SQ_WAIT_INST_ANY"] / agg["SQ_WAVE_CYCLES"], 4))
PY
done
