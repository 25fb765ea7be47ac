#!/bin/bash
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/pmc_sp
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/p1 -- python $ROOT/scripts/ubench/tune_sweep.py 5 2048 256 "" > $OUT/p1.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_WRREQ_WRITE_DRAM_32B_sum --kernel-trace --output-format csv -d $OUT/p2 -- python $ROOT/scripts/ubench/tune_sweep.py 5 2048 256 "" > $OUT/p2.log 2>&1
python - $OUT <<'PY'
import csv, glob, collections, sys
out=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
for f in glob.glob(out+"/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0].replace('void ','').replace('rayrs::','').split('<')[0]
        agg[k][r['Counter_Name']]+=float(r['Counter_Value']); n[k].add(r['Dispatch_Id'])
for k,c in agg.items():
    if not k.startswith(('wf_','sp_')): continue
    simd=c['GRBM_GUI_ACTIVE']/8*1024
    print(k, len(n[k]), 'valu_busy', round(c['SQ_INSTS_VALU']*4/max(simd,1),3), 'VALU G', round(c['SQ_INSTS_VALU']/1e9,2), 'SALU G', round(c['SQ_INSTS_SALU']/1e9,2), 'wait_any/wave_cycles', round(c['SQ_WAIT_ANY']/max(c['SQ_WAVE_CYCLES'],1),3), 'fabric GB', round((c['TCC_EA0_RDREQ_DRAM_32B_sum']+c['TCC_EA0_WRREQ_WRITE_DRAM_32B_sum'])*32/1e9,1))
PY
rm -rf $OUT/p1 $OUT/p2
