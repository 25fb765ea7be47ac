# What memory rates can the shading kernels' access pattern reach on this box?  (GPU box)
# usage: bash scripts/ubench/memory_pattern.sh  -> gpurun_out/memory_pattern.txt
ROOT=${GRAFT_REPO_ROOT:-.}
OUT=$ROOT/gpurun_out
mkdir -p $OUT /tmp/ub
for n in copy_rate pool_rw pool_coop; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/ub/$n $ROOT/scripts/ubench/$n.hip || exit 1
done
{
  echo "## copy_rate: streaming rates with ordinary and non-temporal (nt) accesses, 6 GiB buffers"
  /tmp/ub/copy_rate
  echo "## pool_rw: per-lane reads of whole 192-byte slots + 128-byte write-back, windows compacted as the hit kernel does"
  /tmp/ub/pool_rw
  echo "## pool_coop: the same bytes moved cooperatively (coalesced loads, LDS transpose)"
  /tmp/ub/pool_coop
} > $OUT/memory_pattern.txt 2>&1
cat $OUT/memory_pattern.txt
