ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/calib2
mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o $ROOT/scripts/ubench/fetch_calib $ROOT/scripts/ubench/fetch_calib.hip || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace --output-format csv -d $OUT/a -- $ROOT/scripts/ubench/fetch_calib > $OUT/a.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_32B_sum TCC_BUBBLE_sum TCC_EA0_WRREQ_WRITE_DRAM_32B_sum --kernel-trace --output-format csv -d $OUT/b -- $ROOT/scripts/ubench/fetch_calib > $OUT/b.log 2>&1
python - $OUT <<'PY'
import csv,glob,sys,collections
agg=collections.defaultdict(dict)
for f in glob.glob(sys.argv[1]+"/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "gather" in r["Kernel_Name"]: agg[r["Kernel_Name"][:20]][r["Counter_Name"]]=float(r["Counter_Value"])
for k,v in sorted(agg.items()): print(k, {a:f"{b:.4g}" for a,b in sorted(v.items())})
PY
