#!/bin/bash
# usage (on the GPU box, through gpurun): bash scripts/profile_round.sh <tag> [bench.py flags, e.g. --config 3]
# Produces under gpurun_out/<tag>/:
#   bench.json         the full line `python bench.py [flags]` prints
#   kernel_stats.csv   rocprofv3 --kernel-trace --stats of ONE bench step (same flags)
#   pmc.json           per-kernel PMC sums of the same step, stamped with the source hash and the workload key
#                      bench.py matches them by: fabric bytes (32 B x TCC_EA0_RDREQ_DRAM_32B / WRREQ_WRITE_DRAM_32B,
#                      calibrated by scripts/fetch_calib.sh) and VALU occupancy (SQ_INSTS_VALU x 4 cycles over
#                      GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs).
# Counters are collected in their own rocprofv3 runs with --kernel-trace only.  The library is built before
# any profiled process starts, and every profiled bench runs with --no-build.
# Copy what should be judged to profiles/ as rNN_<tag>_{bench.json,kernel_stats.csv,pmc.json}.
set -e
TAG=${1:-round}; shift || true
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
python -c 'import __graft_entry__ as g; g.build()'
python bench.py --no-build --no-cpu-baseline --no-configs --no-fast --no-secondary --no-roofline "$@" > $OUT/bench.json 2> $OUT/bench.err  # (for the hash and the workload key)
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --no-build --steps 1 --warmup 0 --no-cpu-baseline --no-roofline $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/stats.log 2>&1
cp $OUT/stats/*/*kernel_stats.csv $OUT/kernel_stats.csv
echo "stats done"
rocprofv3 --pmc TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_WRREQ_WRITE_DRAM_32B_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --kernel-trace --output-format csv -d $OUT/pmc_mem -- $CMD > $OUT/pmc_mem.log 2>&1
echo "memory pass done"
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/pmc_sq -- $CMD > $OUT/pmc_sq.log 2>&1
echo "issue pass done"
python - "$OUT" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for which in ("pmc_mem", "pmc_sq"):
    for f in glob.glob(f"{out}/{which}/*/*counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("rayrs::", "").split("<")[0]
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
            disp[k].add((which, row["Dispatch_Id"]))
res = {}
for k, c in agg.items():
    if not k.startswith(("wf_", "lp_", "resolve")): continue
    n = len({d for w, d in disp[k] if w == "pmc_sq"})
    simd_cycles = c["GRBM_GUI_ACTIVE"] / 8 * 1024
    rd, wr = c["TCC_EA0_RDREQ_DRAM_32B_sum"] * 32, c["TCC_EA0_WRREQ_WRITE_DRAM_32B_sum"] * 32
    res[k] = {"launches": n, "fabric_read_bytes": rd, "fabric_write_bytes": wr, "fabric_bytes": rd + wr,
              "read_requests": c["TCC_EA0_RDREQ_sum"], "write_requests": c["TCC_EA0_WRREQ_sum"],
              "valu_wave_instructions": c["SQ_INSTS_VALU"], "salu_wave_instructions": c["SQ_INSTS_SALU"],
              "gpu_cycles_per_xcd": c["GRBM_GUI_ACTIVE"] / 8,
              "valu_busy": round(c["SQ_INSTS_VALU"] * 4 / max(simd_cycles, 1), 4),
              "wave_slot_occupancy_quadcycles": c["SQ_WAVE_CYCLES"], "wait_any": c["SQ_WAIT_ANY"],
              "wait_inst_any": c["SQ_WAIT_INST_ANY"], "busy_cycles": c["SQ_BUSY_CYCLES"]}
bench = json.loads(open(out + "/bench.json").read().strip().splitlines()[-1])
json.dump({"source": "scripts/profile_round.sh: two rocprofv3 --pmc passes (memory side; issue side), --kernel-trace "
                     "only, around `python bench.py --no-build --steps 1 --warmup 0 --no-cpu-baseline --no-roofline`",
           "source_hash": bench["config"]["source_hash"], "workload_key": bench["config"]["workload_key"],
           "workload": bench["config"]["workload"], "kernels": res}, open(out + "/pmc.json", "w"), indent=1)
for k, v in res.items():
    print(k, v["launches"], "fabric GB", round(v["fabric_bytes"] / 1e9, 2), "valu_busy", v["valu_busy"])
PY
rm -rf $OUT/pmc_mem $OUT/pmc_sq $OUT/stats
# the bench line again, now that this tree's counters exist: bench.py fills roofline.traffic / fabric_GBps /
# fp64_valu_busy from profiles/*_pmc.json when its source hash and workload key match
cp $OUT/pmc.json $ROOT/profiles/${TAG}_pmc.json
cd $ROOT
python bench.py --no-build "$@" > $OUT/bench.json 2> $OUT/bench.err
tail -c 200 $OUT/bench.json; echo
