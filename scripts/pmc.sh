#!/bin/bash
# usage: scripts/pmc.sh <tag> <probe-arg>   (run on the GPU box via gpurun)
set -e
TAG=$1; ARG=$2
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
P2="SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM"
P3="TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"
P4="FETCH_SIZE"
P5="WRITE_SIZE GRBM_GUI_ACTIVE"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
  i=$((i+1))
  rocprofv3 --pmc $P --output-format csv -d $OUT/p$i -- python $GRAFT_REPO_ROOT/scripts/perf_probe.py $ARG > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob("$OUT/p*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        if 'trace_kernel' in row['Kernel_Name']:
            agg[row['Counter_Name']] += float(row['Counter_Value']); n[row['Counter_Name']] += 1
for k in sorted(agg): print(f"{k:32s} {agg[k]:.6g}  (dispatches {n[k]})")
PY
