"""The register budget of the hot kernels, from the ISA hipcc emits for gfx950 (no GPU needed).

Occupancy is decided by these numbers and nothing at run time says so when they move: 52 bytes of scratch in the
traversal kernel's loop cost it 45 % (lists by octant), a 97th VGPR its fifth wave per SIMD."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rayrs_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# the flags of rayrs_amd/csrc/Makefile
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-mfma", "-S",
         "--cuda-device-only"]


def kernel_resources(source, tmp_path):
    out = tmp_path / (source + ".s")
    subprocess.run([HIPCC, *FLAGS, "-o", str(out), source], cwd=CSRC, check=True, capture_output=True)
    res = {}
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", out.read_text(), re.S):
        body = m.group(2)
        res[m.group(1)] = (int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", body).group(1)),
                           int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body).group(1)))
    return res


def pick(res, *parts):
    names = [n for n in res if all(p in n for p in parts)]
    assert names, parts
    return {n: res[n] for n in names}


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_streaming_kernels_keep_their_occupancy(tmp_path):
    res = kernel_resources("wavefront.hip", tmp_path)
    # the timed traversal kernels (COUNT = false, EXACT = false): five waves per SIMD, nothing in scratch
    # (template arguments: COMPACT, COUNT, EXACT, PRE)
    for name, (vgpr, scratch) in pick(res, "wf_trav_kernelILb", "ELb0ELb0ELb0EEE").items():
        assert vgpr <= 96 and scratch == 0, (name, vgpr, scratch)
    # the default walk's instances (EXACT: nothing culled, a lane's leaf groups set aside): the same occupancy; at most the
    # stack strip's pointer (read on the deep-stack path only) and two values of the refill branch parked in scratch,
    # never a spill inside an interior or a leaf step (checked in the ISA when the bound was set: every scratch access sits
    # beside a store to the HBM strip or in the refill)
    for name, (vgpr, scratch) in pick(res, "wf_trav_kernelILb", "ELb0ELb1ELb0EEE").items():
        assert vgpr <= 96 and scratch <= 16, (name, vgpr, scratch)
    # ... for pre-tested rays (PRE: scenes with a hot group): the same
    for name, (vgpr, scratch) in pick(res, "wf_trav_kernelILb", "ELb0ELb1ELb1EEE").items():
        assert vgpr <= 96 and scratch <= 32, (name, vgpr, scratch)
    # hit: THREE waves per SIMD since the look-ahead batch no longer lives across Material::evaluate (its slot records are
    # requested behind it: 241 -> 172 registers by itself; the launch bound takes the last four: 32 bytes of scratch,
    # seven accesses in the whole kernel), miss: three and no scratch
    for name, (vgpr, scratch) in pick(res, "wf_hit_kernel").items():
        assert vgpr <= 168 and scratch <= 32, (name, vgpr, scratch)
    for name, (vgpr, scratch) in pick(res, "wf_miss_kernel").items():
        assert vgpr <= 168 and scratch == 0, (name, vgpr, scratch)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_local_pool_kernel_fits_three_workgroups_per_cu(tmp_path):
    """168 registers = three workgroups per CU.  The timed build for the f64 layout -- the reference's sphere scenes,
    configs 1, 2 and 4 -- spills nothing (round 3: 16 registers, 64..80 bytes); the compact one (a handful of
    triangles with f32 vertices) may park one pair."""
    res = kernel_resources("local_pool.hip", tmp_path)
    for name, (vgpr, scratch) in pick(res, "lp_path_kernel").items():
        assert vgpr <= 168, (name, vgpr)
        inst = name.split("lp_path_kernelILb")[1][:8]
        if inst.startswith("0ELb0E"):    # COMPACT = false, COUNT = false
            assert scratch == 0, (name, scratch)
        elif inst.startswith("1ELb0E"):  # COMPACT = true, COUNT = false
            assert scratch <= 16, (name, scratch)
