#!/usr/bin/env python
"""bench.py -- Mray/s of the MI355X radiance integrator on BASELINE.json's headline
configuration (configs[4]: floor + 1,310,720-triangle mesh, 2048x2048, 1024 spp,
50 bounces), on N GPUs of one node.

One step = one full frame: every 8x8 image tile of this rank pushed through the
gfx950 path pipeline (persistent traversal kernel + hit/miss kernels, one round
per bounce; scenes of at most one walk-tree record -- configs 1, 2, 4 -- through the
local-pool kernel instead, every path resident in LDS), the per-pixel resolve, and
(N > 1) one RCCL reduce of the f32x3 framebuffer to rank 0.  Tiles are interleaved over ranks (tile t ->
rank t % N), the scene is replicated, total work is fixed: strong scaling.
`value` = BVH queries of all ranks / max-over-ranks wall time (scene build,
upload and file I/O excluded -- the region the reference times, main.rs:59-100).

Prints ONE JSON line on rank 0 (contract in the task statement) carrying
  roofline     -- the kernel with the largest share of the step (named in `kernel`; every kernel's share and
                  fraction is in `kernels`) against the ceiling that binds it, FP64-rate vector issue: useful
                  lane-operations per launch (the frame's work counters x the operation model below) / its
                  HIP-event duration / (256 CU x 4 SIMD x 16 lanes x 2.4 GHz).  Beside it: algorithmic bytes
                   and, from the PMC passes of scripts/profile_round.sh when they were taken
                  on THIS source tree, fabric traffic and VALU occupancy of that kernel;
  ray_shares   -- what the frame's BVH queries hit (floor / mesh / nothing ...);
  cpu_baseline -- the CPU oracle in reference mode (recursive un-narrowed traversal over a pointer
                  tree, rayrs-lib's algorithm) timed on this box's host cores on a bounded band of the
                  same frame;
  secondary    -- (headline config, N = 1) the same scene from a camera the mesh fills;
  fast         -- (headline config, N = 1) the same frame by rayrs_render_params.fast_traversal, the walk that makes two
                  bets on the reference's arithmetic (include/rayrs_hip.h): its rate beside the headline's, which is
                  the default walk's -- the reference's visit set by construction --, and the two frames' sha256, asserted equal.
"""
import argparse
import glob
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# RCCL shares buffers between the ranks' processes through dmabuf IPC on this pool's driver
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
# FP64 vector peak in lane-operations: 256 CUs x 4 SIMDs x 16 f64 lanes per clock x 2.4 GHz (the guide's
# chip table; MI355X's 78.6 TFLOP/s datasheet figure is this x 2 for fused multiply-adds, which the
# reference's arithmetic -- compiled without contraction on both sides -- never uses).
F64_PEAK_TOPS = 256 * 4 * 16 * 2.4e9 / 1e12

# f64-rate vector instructions one lane needs per unit of traversal work, counted in the kernel source
# (rayrs_amd/csrc/device_path.h) at full lane utilisation; DESIGN.md section 5 derives each number.
# Everything else the kernel issues -- idle lanes of a partly filled wave, ballots, refills, address and
# stack arithmetic, 32-bit selects -- is overhead against this.
OPS_RECORD_COMPACT = 111   # 4 slots x (6 cvt + 6 sub + 6 mul + 6 min/max) + 4 + 4 compares + 1 mul (cull) + 6 (rank)
OPS_RECORD_F64 = 87        # the same without the 24 f32 -> f64 conversions
OPS_RECORD_NO_CULL = 11    # of either: the cull margin's multiply, 4 cull compares, 6 rank compares -- the fast walk only; the
                           # default walk (nothing culled, slots entered in slot order) is not charged with them
OPS_TRIANGLE_COMPACT = 98  # 9 cvt + 9 sub + 2 cross (18) + 4 dot (20) + 3 div x 11 + 1 add + 8 compares
OPS_TRIANGLE_F64 = 89
OPS_SPHERE = 66            # geometry.rs:106-132: 3 sub, 3 dot, sqrt (10), 2 div x 11, 12 others
OPS_PLANE = 26             # geometry.rs:229-271: 1 div x 11, 1 sub, 3 mul, 3 add, 8 compares
OPS_RAY = 54               # 1/d (3 div x 11) + the root box test (21)
# the hot-group phase (device_path.h hot_group_step): wave-uniform f64 data, e1 and e2 from the host
OPS_HOT_GATE = 19          # the group's gating box: 6 sub + 6 mul + 6 min/max + 1 compare (its twelve 32-bit selects are not f64 work)
OPS_HOT_TRIANGLE = 58      # 3 sub + 2 cross (18) + 4 dot (20) + the early rejection's 17 compares, xors and one multiply
OPS_HOT_DIVISIONS = 42     # on top, for a triangle some lane of the wave was not settled on: 3 div x 11 + 1 add + 8 compares

# REFERENCE ARITHMETIC ONLY (roofline.frac_ref_flops): the f64 additions, multiplications, divisions, square roots,
# min/max and compares that BvhTree::intersect and the shapes' intersect functions perform on what the walk visits -- a
# division or a square root counts ONE, nothing is charged for conversions, selects, the shortcut's tests or bookkeeping.
# Holds against the datasheet's vector FP64 rate without FMA (39.3 T/s); by construction below `frac`.
REF_BOX = 19               # geometry.rs:458-513 with 1 / d at hand: 6 sub, 6 mul, 6 min/max, 1 compare
REF_RECORD = 4 * REF_BOX   # the walk's records hold four boxes (an unused slot's inverted box is tested like any other)
REF_TRIANGLE = 51          # geometry.rs:359-375: t = o - p1 (3), two cross products (18), four dot products (20), 3 div, u + v, 4 compares; bvh.rs:406: 2
REF_TRIANGLE_SETTLED = 41  # ... of a hot-group triangle rejected before the divisions: the subtraction, the cross and dot products
REF_SPHERE = 33            # geometry.rs:106-132: 3 sub, three dot products (15), 2 mul, 4 for the discriminant, sqrt, 2 x (add, mul, div), 2 compares
REF_PLANE = 14             # geometry.rs:229-271: 1 sub, 1 div, 3 mul + 3 add, 4 compares; bvh.rs:406: 2
REF_RAY = 3 + REF_BOX      # 1 / d and the root Node's box


# Vector instructions one evaluation of a shading unit costs when the unit is compiled alone and all 64 lanes work:
# SQ_INSTS_VALU x 64 / evaluations of rayrs_test_material / rayrs_test_background on 2 M random tuples each, minus
# the test kernel's own loads and stores (scripts/op_model.sh -> profiles/rNN_op_model.json, stamped with the hash of
# the shading sources it was measured on; load_op_model() below takes the newest one and says whether it is stale).
# The dielectrics are evaluated with half of the hits from inside the medium; their arms branch (reflect / refract /
# total internal reflection), and a wave runs every branch one of its lanes takes -- that is part of what the arm costs.
# (These literals are round 3's measurement: the fallback when no profile is found, flagged as such in the line.)
OPS_MATERIAL = {
    "lambertian": 317, "reflect": 30, "refract": 136, "glass": 248, "cook_torrance": 831,
    "cook_torrance_refract": 1120, "cook_torrance_glass": 2120, "plastic": 1187, "no_reflect": 0,
}
OPS_BACKGROUND = 495       # Scene::background: unit, atan2, acos, the 2x2 footprint, 12 products (lib.rs:254-285)
# counted in the source (rayrs_amd/csrc/wavefront.hip, device_path.h), like the traversal units:
OPS_HIT_FIXED = 150        # position 6, view 32, normal ~32, emission and throughput 9, max 2, roulette draw ~30, 3 div x 11, compares
OPS_SAMPLE = 180           # a new sample: path key ~40, two draws ~60, primary ray ~25, 1/d and the root box test 54
MATERIAL_UNIT = ["lambertian", "reflect", "refract", "glass", "cook_torrance", "cook_torrance_refract",
                 "cook_torrance_glass", "plastic", "no_reflect"]   # by RAYRS_MAT_*
OP_MODEL_UNIT = {"lambertian": "lambertian", "reflect": "reflect", "refract": "refract", "glass": "glass",
                 "cook_torrance": "cook_torrance_metal_rough", "cook_torrance_refract": "cook_torrance_refract",
                 "cook_torrance_glass": "cook_torrance_glass", "plastic": "plastic"}
OP_MODEL = {"file": None, "stale": True, "shading_source_hash": None,
            "note": "no profiles/*_op_model.json found: round 3's literals (bench.py OPS_MATERIAL)"}


def shading_source_hash():
    """sha256 over the sources the shading units are compiled from (what scripts/op_model.sh measures)."""
    h = hashlib.sha256()
    for f in ("include/rayrs_numeric.h", "rayrs_amd/csrc/device_path.h", "rayrs_amd/csrc/kernels.hip", "rayrs_amd/csrc/layout.h"):
        h.update(f.encode())
        h.update(open(os.path.join(ROOT, f), "rb").read())
    return h.hexdigest()[:16]


def load_op_model():
    """The newest profiles/*_op_model.json becomes OPS_MATERIAL / OPS_BACKGROUND; OP_MODEL says which file and whether
    it was measured on the shading sources of this tree (a two-rounds-old model went unnoticed)."""
    global OPS_BACKGROUND
    want = shading_source_hash()
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_op_model.json"))):
        try:
            j = json.load(open(path))
        except (OSError, ValueError):
            continue
        if "units" in j and (best is None or j.get("shading_source_hash") == want or best[1].get("shading_source_hash") != want):
            best = (path, j)
    if best is None:
        return
    path, j = best
    u = j["units"]
    for unit, key in OP_MODEL_UNIT.items():
        if key in u:
            OPS_MATERIAL[unit] = int(round(u[key]["valu_per_evaluation_net"]))
    if "background" in u:  # (minus the test kernel's own loads and stores, like the material units)
        OPS_BACKGROUND = int(round(u["background"]["valu_per_evaluation"] - u["none"]["valu_per_evaluation"]))
    stale = j.get("shading_source_hash") != want
    OP_MODEL.update({"file": os.path.relpath(path, ROOT), "stale": stale, "shading_source_hash": j.get("shading_source_hash"),
                     "this_tree": want,
                     "note": ("measured on other shading sources than this tree's: re-run scripts/op_model.sh" if stale
                              else "measured on this tree's shading sources")})


def hot_group_kinds(scene, info, objs):
    """(triangles, spheres, rectangles) among the primitives of the scene's hot group."""
    if not info["hot_count"]:
        return 0, 0, 0
    from rayrs_amd.api import flatten_objects
    spans, at = [], 0   # (first object index, count, kind) in insertion order: a "mesh" is one object per triangle
    for o in flatten_objects(objs):
        n = len(o.idx) if o.kind == "mesh" else 1
        spans.append((at, n, "triangle" if o.kind == "mesh" else o.kind))
        at += n
    pp = scene.export_bvh()[2]
    kinds = []
    for k in range(info["hot_count"]):
        i = int(pp[info["hot_first"] + k])
        kinds.append(next(kind for first, n, kind in spans if first <= i < first + n))
    return kinds.count("triangle"), kinds.count("sphere"), kinds.count("plane")


def hot_counts(stats, hot_kinds):
    """(triangle, sphere, rectangle) tests of the frame made on the hot group, by the kernels that make the rays."""
    if not stats.get("hot_group"):
        return 0, 0, 0
    entered = stats["hot_prim_tests"] // max(sum(hot_kinds), 1)
    return tuple(entered * k for k in hot_kinds)


def traversal_ops(stats, info, hot_kinds=(0, 0, 0)):
    """Useful f64-rate lane operations of the BVH walks the TRAVERSAL kernel (or the local-pool kernel) makes: records
    entered, primitives tested.  On a scene with a hot group the kernels that make the rays test that group and the walk
    tree's first record (pretest_ops): those tests are not in here."""
    rec = OPS_RECORD_COMPACT if info["compact"] else OPS_RECORD_F64
    if stats.get("exact_walk"):  # (the local-pool kernel's loop over a record's gates neither culls nor ranks either)
        rec -= OPS_RECORD_NO_CULL
    tri = OPS_TRIANGLE_COMPACT if info["compact"] else OPS_TRIANGLE_F64
    h_tri, h_sph, h_pl = hot_counts(stats, hot_kinds)
    return ((stats["interior_visits"] - stats.get("pre_root_records", 0)) * rec + (stats["tri_tests"] - h_tri) * tri
            + (stats["sphere_tests"] - h_sph) * OPS_SPHERE + (stats["plane_tests"] - h_pl) * OPS_PLANE)


def traversal_rays(stats):
    """Rays the traversal kernel takes from the pool: not the primary rays that miss the root box (direct_rays: the kernel
    that makes them finishes their sample), not the queries answered by the pre-test (pre_rays)."""
    return stats["rays"] - stats.get("direct_rays", 0) - stats.get("pre_rays", 0)


def pretest_ops(stats, info, hot_kinds=(0, 0, 0)):
    """Useful lane operations of the pre-test (wavefront.hip finish_rays) in the kernels that make the rays: 1 / d and the
    root box for the bounced rays (a new sample's are in OPS_SAMPLE), the hot group's gate and primitives, the walk tree's
    first record for every ray that enters the root box -- all on wave-uniform f64 data."""
    if not stats.get("hot_group"):
        return 0
    h_tri, h_sph, h_pl = hot_counts(stats, hot_kinds)
    bounced = stats["rays"] - stats["paths"]
    return (bounced * OPS_RAY + stats["hot_lane"] * (OPS_HOT_GATE + 4 * OPS_HOT_GATE) + h_tri * OPS_HOT_TRIANGLE
            + stats["hot_tri_divided"] * OPS_HOT_DIVISIONS + h_sph * OPS_SPHERE + h_pl * OPS_PLANE)


def reference_flops(stats, info, hot_kinds=(0, 0, 0)):
    """roofline.frac_ref_flops' numerator for the traversal kernel: reference arithmetic only (REF_* above), of the
    records and primitives IT visits."""
    h_tri, h_sph, h_pl = hot_counts(stats, hot_kinds)
    return ((stats["interior_visits"] - stats.get("pre_root_records", 0)) * REF_RECORD + (stats["tri_tests"] - h_tri) * REF_TRIANGLE
            + (stats["sphere_tests"] - h_sph) * REF_SPHERE + (stats["plane_tests"] - h_pl) * REF_PLANE
            + traversal_rays(stats) * (3 if stats.get("hot_group") else REF_RAY))


def useful_f64_ops(stats, info, hot_kinds=(0, 0, 0)):
    """Of the traversal kernel: the walks plus the setup of the rays it took (pre-tested rays: 1 / d only -- their root box
    was the pre-test's)."""
    return traversal_ops(stats, info, hot_kinds) + traversal_rays(stats) * (OPS_RAY - 21 if stats.get("hot_group") else OPS_RAY)


def layout_conversion_ops(stats, info, hot_kinds=(0, 0, 0)):
    """The f32 -> f64 conversions of the compact layout among traversal_ops(): they exist because of the layout, not
    because of the reference's arithmetic (24 per record, 9 per triangle read from its record)."""
    if not info["compact"]:
        return 0
    h_tri = hot_counts(stats, hot_kinds)[0]
    return ((stats["interior_visits"] - stats.get("pre_root_records", 0)) * (OPS_RECORD_COMPACT - OPS_RECORD_F64)
            + (stats["tri_tests"] - h_tri) * (OPS_TRIANGLE_COMPACT - OPS_TRIANGLE_F64))


def surface_units(objs):
    """Material unit of every (Material, Emission) row, in insertion order (what surface_hits is indexed by)."""
    from rayrs_amd.api import flatten_objects
    rows = []
    for o in flatten_objects(objs):
        key = (o.mat, o.emission)
        if key not in rows:
            rows.append(key)
    return [MATERIAL_UNIT[m.kind] for m, _ in rows]


def shading_ops(stats, units):
    """(hit side, miss side, new samples) useful lane operations of a frame from its counters: closest hits per
    surface row x that row's material unit; escaped paths x Scene::background; samples started."""
    hits = stats["surface_hits"]
    ops_hit = 0
    for k, n in enumerate(hits):
        unit = units[k] if k < len(units) else units[-1]
        ops_hit += n * (OPS_MATERIAL[unit] + OPS_HIT_FIXED)
    return ops_hit, stats["escaped_paths"] * OPS_BACKGROUND, stats["paths"] * OPS_SAMPLE


def algorithmic_bytes(stats, info):
    """bytes the traversal kernel must move on the flattened layout:
    one record per interior visit and per primitive test (of ITS walks: the pre-test's data is wave-uniform, a few
    hundred bytes per wave), plus, per BVH query it takes, the ray it reads from the path pool (origin + direction, 48 B; a
    pre-tested ray's closest hit so far, 12 B), the result it writes back (t + primitive, 12 B) and the slot's state
    byte (read + write)."""
    prims = stats["tri_tests"] + stats["sphere_tests"] + stats["plane_tests"] - stats.get("hot_prim_tests", 0)
    return ((stats["interior_visits"] - stats.get("pre_root_records", 0)) * info["node_bytes"] + prims * info["prim_bytes"]
            + traversal_rays(stats) * (48 + 12 + 2 + (12 if stats.get("hot_group") else 0)))


def source_hash():
    """sha256 over the sources the library is built from: PMC profiles are only valid for the tree they
    were collected on (scripts/profile_round.sh stamps them with this)."""
    files = sorted(glob.glob(os.path.join(ROOT, "rayrs_amd", "csrc", "*.hip"))
                   + glob.glob(os.path.join(ROOT, "rayrs_amd", "csrc", "*.h"))
                   + glob.glob(os.path.join(ROOT, "rayrs_amd", "csrc", "*.hpp"))
                   + glob.glob(os.path.join(ROOT, "rayrs_amd", "csrc", "*.cpp"))
                   + glob.glob(os.path.join(ROOT, "include", "*.h")))
    h = hashlib.sha256()
    for f in files:
        if f.endswith("rayrs_cli.cpp"):
            continue
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def find_pmc_profile(workload_key):
    """The newest profiles/*_pmc.json taken on this source tree for this workload, or None."""
    want = source_hash()
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json"))):
        try:
            j = json.load(open(path))
        except (OSError, ValueError):
            continue
        if j.get("source_hash") == want and j.get("workload_key") == workload_key:
            best = (os.path.relpath(path, ROOT), j)
    return best


def effective_cores():
    """Host cores this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def surface_names(config, n_surfaces):
    """Names of the (Material, Emission) rows in insertion order, from rayrs_amd/scenes.py's construction."""
    if config in (3, 5):
        names = ["floor", "mesh", "light"]
    else:
        names = ["floor"] + [f"sphere{i}" for i in range(7)]
    return names[:n_surfaces] if n_surfaces <= len(names) else names + ["other"] * (n_surfaces - len(names))


def ray_shares(cst, config, n_surfaces):
    names = surface_names(config, min(n_surfaces, 8))
    rays = max(cst["rays"], 1)
    out = {}
    for k, name in enumerate(names):
        key = name if not name.startswith("sphere") else "spheres"
        out[key] = out.get(key, 0.0) + cst["surface_hits"][k] / rays
    out["nothing"] = (cst["rays"] - sum(cst["surface_hits"])) / rays
    return {k: round(v, 4) for k, v in out.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=5, help="BASELINE.json configs[n-1]; 5 is the headline")
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel (development only)")
    ap.add_argument("--res", type=int, default=0, help="override resolution (development only)")
    ap.add_argument("--camera", default="reference", choices=["reference", "close"],
                    help="'close' = the mesh-filling camera of the secondary line (mesh configs only)")
    ap.add_argument("--sample-chunk", type=int, default=4,
                    help="smallest sample chunk to use (the library doubles it until the whole frame has at most "
                         "2^30 items); small chunks keep the end of a frame, and of a tile share, short")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--fast-traversal", action="store_true",
                    help="render the timed frames with rayrs_render_params.fast_traversal (the two bets) instead of the "
                         "default walk; the default line carries that figure in its `fast` block anyway")
    ap.add_argument("--no-fast", action="store_true", help="skip the `fast` block")
    ap.add_argument("--no-configs", action="store_true", help="skip the `configs` block (configs[0..3] beside the headline)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend; 'gloo' (host reduce) is for "
                    "rehearsing the N>1 path on a box with fewer GPUs than ranks")
    ap.add_argument("--device", type=int, default=-1, help="force the HIP device of every rank (rehearsal only)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed and run the framebuffer reduce also with one rank (under "
                         "torch.distributed.run --nproc-per-node=1): the RCCL code path of an N-GPU run on a one-GPU box")
    ap.add_argument("--no-stagger", action="store_true", help="development only: start all flights at once")
    ap.add_argument("--share-of", type=int, default=0,
                    help="development only: render the tile share rank 0 of N would (what one GPU of N does), on one GPU")
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="frames rendered at a time, each by its own host thread on its own clone of the scene (own path "
                         "pool) and HIP stream: the end of a frame -- a thinning pool, small launches -- overlaps the start "
                         "of the next (and a rank's RCCL reduce the next frame's rendering).  1 = one frame after the other; "
                         "0 = 3 when the frame is shared among GPUs (the end of a one-eighth share is 12 %% of it: 170 -> 158 -> 156 ms "
                         "per frame with 2 and 3; a quarter 322 -> 310 -> 305; a half 617 -> 614 -> 612), 1 on one GPU (2 %% of a "
                         "whole frame: measured +-0, for twice the pool memory)")
    ap.add_argument("--no-build", action="store_true",
                    help="do not run make: required under rocprofv3 (a profiled process must not spawn the compiler); "
                         "fails if the library is older than its sources")
    args = ap.parse_args()

    import numpy as np
    import torch

    load_op_model()
    import __graft_entry__ as graft
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.device >= 0:
        local_rank = args.device
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.no_build:
        graft.check_built()
    elif rank == 0:
        graft.build()

    use_dist = world > 1 or args.force_dist
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
        dist.barrier()
    else:
        torch.cuda.set_device(local_rank)

    import rayrs_amd
    from rayrs_amd import procedural, scenes, tiles

    import tempfile
    import threading
    import types
    dev = torch.device("cuda", local_rank)
    stream = torch.cuda.current_stream(dev)
    hdri = procedural.make_hdri(1024, 512)
    RES = {1: 256, 2: 1024, 3: 1024, 4: 2048, 5: 2048}

    def make_context(config, n_flights_want, spp_override=0, res_override=0, camera="reference"):
        """Scene, camera and flights of BASELINE.json configs[config - 1]."""
        cx = types.SimpleNamespace(config=config)
        # the mesh takes the route a scanned model would: written as a binary PLY (by rank 0 of the node; the other ranks
        # wait for it), read back by every rank through the library's loader
        ply_path = None
        if config in (3, 5):
            ply_path = os.path.join(tempfile.gettempdir(), f"rayrs_bench_mesh_config{config}_{os.environ.get('MASTER_PORT', 'solo')}_{os.getuid()}.ply")
            if use_dist:
                if rank == 0:
                    scenes.write_config_ply(config, ply_path)
                dist.barrier()
                cam_args, objs, heur, spp, max_bounces = scenes.config(config, ply_path=ply_path, ply_exists=True)
            else:
                cam_args, objs, heur, spp, max_bounces = scenes.config(config, ply_path=ply_path)
        else:
            cam_args, objs, heur, spp, max_bounces = scenes.config(config)
        cx.reduced = False
        if spp_override:
            spp, cx.reduced = spp_override, True
        W0 = H0 = RES[config]
        if res_override:
            W0 = H0 = res_override
            cx.reduced = True
        if camera == "close":
            if config not in (3, 5):
                raise SystemExit("--camera close is for the mesh configurations (3, 5)")
            cam_args = scenes.MESH_CLOSE_CAM
        cx.cam_args = scenes.camera_for_resolution(cam_args, W0, H0)
        cx.W0, cx.H0 = W0, H0
        cx.objs, cx.heur, cx.spp, cx.max_bounces = objs, heur, spp, max_bounces
        t0 = time.time()
        cx.scene = rayrs_amd.Scene(objs, 1e-6, 1e6, heur, hdri, device=local_rank)
        cx.build_s = time.time() - t0
        cx.info = cx.scene.info()
        # frames in flight: flight 0 is the scene itself on the current stream; the others are clones (same records, own
        # pool) on streams of their own
        cx.n_flights = max(1, n_flights_want)
        # what a rank is about to hold in HBM, before it allocates anything of it (an N-rank out-of-memory failure must be
        # readable from the log): the scene + per flight a path pool and the item sums, as abi.cpp rayrs_render_launch sizes them
        cam = rayrs_amd.Camera(*cx.cam_args)
        chunk = rayrs_amd.frame_sample_chunk(cam.x_pixels(), cam.y_pixels(), spp, args.sample_chunk) or spp
        n_ranks = args.share_of if args.share_of else world
        tiles_total = ((cam.x_pixels() + 7) // 8) * ((cam.y_pixels() + 7) // 8)
        local_tiles = (tiles_total - (0 if args.share_of else rank) + n_ranks - 1) // n_ranks
        items = local_tiles * 64 * ((spp + chunk - 1) // chunk)
        if cx.info["local_pool"]:
            pool_b, sums_b = 0, min(items, 1 << 27) * 24
        else:
            slots = min(items, 1 << 28, max(local_tiles * 64 * spp // 12, 1 << 20))
            pool_b, sums_b = ((slots + 1023) & ~1023) * (128 + 32 + 1), items * 24
        cx.hbm_bytes = int(cx.info["device_bytes"] + cx.n_flights * (pool_b + sums_b))
        print(f"bench.py rank {rank}/{world} config {config}: about to hold scene {cx.info['device_bytes'] / 1e6:.0f} MB + {cx.n_flights} "
              f"flight(s) x (path pool {pool_b / 1e9:.2f} GB + item sums {sums_b / 1e9:.2f} GB) = {cx.hbm_bytes / 1e9:.2f} GB on HIP device "
              f"{local_rank}", file=sys.stderr, flush=True)
        # (with more than one flight none of them uses the null stream, whose launches order against every other stream's)
        cx.flights = [{"scene": cx.scene, "stream": stream if cx.n_flights == 1 else torch.cuda.Stream(dev)}]
        for _ in range(cx.n_flights - 1):
            cx.flights.append({"scene": cx.scene.clone_to_device(local_rank), "stream": torch.cuda.Stream(dev)})
        return cx

    if args.share_of and world != 1:
        raise SystemExit("--share-of is a one-GPU development option")
    n_flights = args.frames_in_flight if args.frames_in_flight > 0 else (3 if (world > 1 or args.share_of) else 1)
    n_flights = max(1, min(n_flights, args.steps))
    cx = make_context(args.config, n_flights, args.spp, args.res, args.camera)
    reduced = cx.reduced or bool(args.share_of)
    cam_args, objs, heur, spp, max_bounces, info, build_s = cx.cam_args, cx.objs, cx.heur, cx.spp, cx.max_bounces, cx.info, cx.build_s
    W0, H0 = cx.W0, cx.H0

    def workload_key(cx, W, H, chunk, camera=None):
        return f"config{cx.config}_{camera or args.camera}_{W}x{H}_{cx.spp}spp_chunk{chunk}_world{world}"

    def measure(cx, cam_args, steps, warmup, want_roofline, fast=False):
        """Times `steps` frames of the scene in `cx` from this camera; returns the pieces of the JSON line."""
        scene, flights, n_flights, objs, info, spp, max_bounces = cx.scene, cx.flights, cx.n_flights, cx.objs, cx.info, cx.spp, cx.max_bounces
        cam = rayrs_amd.Camera(*cam_args)
        H, W = cam.y_pixels(), cam.x_pixels()
        # a pixel's samples are summed in chunks; the chunk comes from the WHOLE frame (rayrs_frame_sample_chunk:
        # at most 2^30 (pixel, chunk) items), never from the rank count, so every N renders the same bits
        chunk = rayrs_amd.frame_sample_chunk(W, H, spp, args.sample_chunk)
        F = max(1, min(n_flights, steps))
        fbs = [torch.zeros((H, W, 3), dtype=torch.float32, device=dev) for _ in range(F)]
        params = rayrs_amd.make_params(spp, max_bounces, seed=0x5EED, sample_chunk=chunk, tile_rank=rank,
                                       tile_ranks=args.share_of if args.share_of else world, fast_traversal=fast)
        turn = {"n": 0, "failed": None}
        cv = threading.Condition()

        def my_turn(k):  # collectives are issued in frame order, whichever thread renders the frame
            with cv:
                while turn["n"] != k:
                    if turn["failed"] is not None:
                        raise RuntimeError("another flight failed")
                    cv.wait(timeout=1.0)

        def pass_turn():
            with cv:
                turn["n"] += 1
                cv.notify_all()

        def frames(n_frames, stagger_s=0.0, nf=None):
            """Renders n_frames frames, frame k on flight k % nf (default: all F flights); returns their stats in frame
            order.  Flight i starts i * stagger_s late, so that the flights' frames end at different times."""
            stats = [None] * n_frames
            turn["n"], turn["failed"] = 0, None
            nf = max(1, min(nf or F, F, n_frames))

            def flight(i):
                try:
                    torch.cuda.set_device(dev)
                    f, fb = flights[i], fbs[i]
                    if i and stagger_s > 0.0:
                        time.sleep(i * stagger_s)
                    with torch.cuda.stream(f["stream"]):
                        for k in range(i, n_frames, nf):
                            fb.zero_()
                            rayrs_amd.render_launch(f["scene"], cam, params, fb.data_ptr(), f["stream"].cuda_stream)
                            if use_dist and args.backend == "nccl":
                                my_turn(k)
                                tiles.reduce_framebuffer(fb, dst=0)   # RCCL over xGMI, ordered after the render on this stream
                                pass_turn()
                            st = rayrs_amd.render_finish(f["scene"])
                            if use_dist and args.backend != "nccl":   # rehearsal: reduce through host memory
                                my_turn(k)
                                host = fb.cpu()
                                tiles.reduce_framebuffer(host, dst=0)
                                fb.copy_(host)
                                pass_turn()
                            stats[k] = st
                except BaseException as e:  # noqa: BLE001 -- handed to the main thread below
                    with cv:
                        turn["failed"] = e
                        cv.notify_all()

            if nf == 1:
                flight(0)
            else:
                th = [threading.Thread(target=flight, args=(i,)) for i in range(nf)]
                for t in th:
                    t.start()
                for t in th:
                    t.join()
            if turn["failed"] is not None:
                if use_dist:
                    # a flight that died before its collective leaves the other ranks waiting in the reduce for ever:
                    # end the process (torch.distributed.run then ends the job) instead of raising into a hung group
                    import traceback
                    traceback.print_exception(type(turn["failed"]), turn["failed"], turn["failed"].__traceback__)
                    sys.stderr.flush()
                    os._exit(3)
                raise turn["failed"]
            return stats

        def fence():
            if use_dist:
                dist.barrier()
            torch.cuda.synchronize(dev)

        st = None
        # every flight renders at least one untimed frame: a clone's first frame allocates its path pool, stack strip
        # and round events (with --warmup 1 and two flights the second one did that inside the timed region)
        n_warm = max(warmup, F) if (warmup and F > 1) else warmup
        if n_warm:
            st = frames(n_warm)[-1]
        fence()
        t_begin = time.perf_counter()
        timed = frames(steps, stagger_s=(st["trace_ms"] * 1e-3 / F) if (st is not None and F > 1 and not args.no_stagger) else 0.0)
        fence()
        elapsed = time.perf_counter() - t_begin
        st = timed[-1]
        fb = fbs[(steps - 1) % F]
        rays = sum(t["rays"] for t in timed)
        # frames in flight make ms_per_step a throughput figure; beside it, the same frames one after the other (one
        # flight: render, reduce, next frame), so that an N-GPU line can be read against the 1-GPU line like for like
        unpipelined_s = None
        if F > 1:
            fence()
            t_u = time.perf_counter()
            frames(2, nf=1)
            fence()
            unpipelined_s = (time.perf_counter() - t_u) / 2.0
        alone_ms = None
        if F > 1 and want_roofline:
            # the kernels' own durations: one more frame, rendered alone (with frames in flight the kernels of two
            # frames share the GPU and a launch's duration is not its own)
            fb_alone = torch.zeros_like(fbs[0])
            rayrs_amd.render_launch(scene, cam, params, fb_alone.data_ptr(), stream.cuda_stream)
            one = rayrs_amd.render_finish(scene)
            del fb_alone
            timed_for_kernels = [one]
            alone_ms = one["trace_ms"]
        else:
            timed_for_kernels = timed
        kernel_ms = [t["kernel_ms"] for t in timed_for_kernels]
        hit_ms = [t["hit_ms"] for t in timed_for_kernels]
        miss_ms = [t["miss_ms"] for t in timed_for_kernels]

        tot = torch.tensor([float(rays), elapsed, unpipelined_s or 0.0], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        if use_dist:
            r = tot[:1].clone()
            e = tot[1:].clone()
            dist.all_reduce(r, op=dist.ReduceOp.SUM)
            dist.all_reduce(e, op=dist.ReduceOp.MAX)
            total_rays, max_elapsed = float(r.item()), float(e[0].item())
            if unpipelined_s is not None:
                unpipelined_s = float(e[1].item())
        else:
            total_rays, max_elapsed = float(rays), elapsed
        checksum, fb_sha = 0.0, None
        if rank == 0:  # the assembled frame: identical bits for every rank count (tests/test_gpu_multi_process.py)
            checksum = float(fb.double().sum().item())
            fb_sha = hashlib.sha256(fb.cpu().numpy().tobytes()).hexdigest()

        roofline, shares = None, None
        if want_roofline:
            # same launch once more with the work counters compiled in (untimed): the work of a launch is a
            # pure function of (scene, seed), so the counts apply to the timed launches exactly
            pc = rayrs_amd.make_params(spp, max_bounces, seed=0x5EED, sample_chunk=chunk, tile_rank=rank,
                                       tile_ranks=args.share_of if args.share_of else world, count_work=True, fast_traversal=fast)
            fb2 = torch.zeros_like(fb)
            rayrs_amd.render_launch(scene, cam, pc, fb2.data_ptr(), stream.cuda_stream)
            cst = rayrs_amd.render_finish(scene)
            assert cst["rays"] == st["rays"], "counting launch traced a different frame"
            shares = ray_shares(cst, cx.config, info["n_surfaces"])
            launches = st["kernel_launches"]                          # traversal launches = path rounds (local pool: segments)
            step_ms = max_elapsed / steps * 1e3
            share_base_ms = alone_ms if alone_ms is not None else step_ms
            n_st = len(kernel_ms)
            units = surface_units(objs)
            hot_kinds = hot_group_kinds(scene, info, objs)
            ops_hit, ops_miss, ops_gen = shading_ops(cst, units)
            paths_end_in_miss = cst["escaped_paths"]
            share_miss = paths_end_in_miss / max(cst["paths"], 1)
            # every kernel of the frame: its HIP-event time per step (on the render stream) and the useful lane
            # operations it is charged with; the roofline is the one with the largest share of the step
            if st["local_pool"]:
                kern = {"lp_path_kernel": {"ms": sum(kernel_ms) / n_st,
                                           "ops": traversal_ops(cst, info, hot_kinds) + cst["rays"] * OPS_RAY + ops_hit + ops_miss + ops_gen}}
            else:
                # the pre-test of new rays is the work of the kernel that makes them: bounced rays are the hit kernel's, a new
                # sample's primary ray belongs to the kernel its path ended in
                ops_pre = pretest_ops(cst, info, hot_kinds)
                bounced = cst["rays"] - cst["paths"]
                pre_hit = ops_pre * (bounced + cst["paths"] * (1.0 - share_miss)) / max(cst["rays"], 1)
                kern = {"wf_trav_kernel": {"ms": sum(kernel_ms) / n_st, "ops": useful_f64_ops(cst, info, hot_kinds)},
                        "wf_hit_kernel": {"ms": sum(hit_ms) / n_st, "ops": ops_hit + ops_gen * (1.0 - share_miss) + pre_hit},
                        "wf_miss_kernel": {"ms": sum(miss_ms) / n_st, "ops": ops_miss + ops_gen * share_miss + ops_pre - pre_hit}}
            for k in kern.values():
                k["share_of_step"] = round(k["ms"] / share_base_ms, 3)
                k["achieved_Tops"] = round(k["ops"] / max(k["ms"], 1e-9) / 1e9, 3)
                k["frac"] = round(k["ops"] / max(k["ms"], 1e-9) / 1e9 / F64_PEAK_TOPS, 4)
                k["ms"] = round(k["ms"], 3)
                k["ops"] = int(k["ops"])
            dom = max(kern, key=lambda n: kern[n]["ms"])
            dom_ms = kern[dom]["ms"]
            avg_ms = dom_ms / launches
            ops = kern[dom]["ops"]
            achieved = ops / (dom_ms * 1e-3) / 1e12
            # the same without the f32 -> f64 conversions of the compact layout (they are in every traversal figure)
            conv = layout_conversion_ops(cst, info, hot_kinds)
            # reference arithmetic only, a division = 1 (the traversal kernel; shading units are priced in instructions)
            frac_ref = None
            if dom == "wf_trav_kernel":
                frac_ref = round(reference_flops(cst, info, hot_kinds) / (dom_ms * 1e-3) / 1e12 / F64_PEAK_TOPS, 4)
            frac_no_conv = (ops - conv) / (dom_ms * 1e-3) / 1e12 / F64_PEAK_TOPS
            abytes = algorithmic_bytes(cst, info)
            prims = cst["tri_tests"] + cst["sphere_tests"] + cst["plane_tests"]
            pmc = find_pmc_profile(workload_key(cx, W, H, chunk))
            traffic = fabric_gbs = valu_busy = valu_per_ray = valu_per_trav_ray = pmc_src = None
            step_fabric = None
            if pmc is not None:
                pmc_src, pj = pmc
                k = pj["kernels"].get(dom)
                if k and k["launches"] == launches:
                    traffic = int(k["fabric_bytes"] / launches)
                    fabric_gbs = round(k["fabric_bytes"] / (dom_ms * 1e-3) / 1e9, 1)
                    valu_busy = k.get("valu_busy")
                    valu_per_ray = round(k["valu_wave_instructions"] / max(cst["rays"], 1), 2)
                    if dom == "wf_trav_kernel":  # (per ray that goes through the kernel: with the pre-test most do not)
                        valu_per_trav_ray = round(k["valu_wave_instructions"] / max(traversal_rays(cst), 1), 2)
                    step_fabric = sum(kk["fabric_bytes"] for kk in pj["kernels"].values())  # every kernel of ONE frame
            # the whole step: every kernel's useful operations, and (from the PMC passes) all fabric traffic, over the
            # step's wall time -- what the one-kernel-after-the-other schedule makes of the chip as a whole
            step_ops = sum(k["ops"] for k in kern.values())
            whole_step = {
                "ms": round(step_ms, 2), "useful_ops": int(step_ops),
                "achieved_Tlaneops": round(step_ops / (step_ms * 1e-3) / 1e12, 3),
                "frac_of_fp64_issue_peak": round(step_ops / (step_ms * 1e-3) / 1e12 / F64_PEAK_TOPS, 4),
                "fabric_bytes": None if step_fabric is None else int(step_fabric),
                "fabric_GBps": None if step_fabric is None else round(step_fabric / (step_ms * 1e-3) / 1e9, 1),
                "frac_of_hbm_peak": None if step_fabric is None else round(step_fabric / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "kernel_ms_sum": round(sum(k["ms"] for k in kern.values()), 2),
            }
            if st["local_pool"] == 1:
                util = {"all phases": round(cst["step_lane"] / max(cst["step_wave"], 1), 3)}
                tk = max(cst["interior_ticks"] + cst["leaf_ticks"] + cst["refill_ticks"], 1)
                util["wave time: intersect / shade / generate+background"] = [
                    round(cst["interior_ticks"] / tk, 3), round(cst["leaf_ticks"] / tk, 3), round(cst["refill_ticks"] / tk, 3)]
            else:
                util = {"interior": round(cst["step_lane"] / max(cst["step_wave"], 1), 3),
                        "leaf": round(cst["inner_wave"] / max(cst["leaf_wave"], 1), 3)}
            roofline = {
                "bound": "fp64_valu", "achieved": round(achieved, 3), "peak": round(F64_PEAK_TOPS, 2),
                "unit": "Tlane-op/s", "frac": round(achieved / F64_PEAK_TOPS, 4), "traffic": traffic,
                "frac_without_layout_conversions": round(frac_no_conv, 4),
                "frac_ref_flops": frac_ref, "op_model": dict(OP_MODEL),
                "step": whole_step,
                "kernel": dom, "launches_per_step": int(launches), "kernel_ms": round(avg_ms, 4),
                "kernel_times_from": "the timed steps" if F == 1 else
                                     "one more frame rendered alone behind the timed region (frames in flight share the GPU: "
                                     "a launch's duration there is not its own); shares are of that frame's duration",
                "kernel_share_of_step": round(dom_ms / share_base_ms, 3),
                "definition": "the kernel with the largest share of the step: useful lane operations it is charged with / its "
                              "HIP-event time / (256 CU x 4 SIMD x 16 lanes x 2.4 GHz = vector issue rate of f64-width "
                              "instructions; no FMA credit: the reference's arithmetic is unfused).  Traversal units (record, "
                              "primitive, ray) are f64 operations counted in device_path.h (plus the compact layout's f32 -> "
                              "f64 conversions: frac_without_layout_conversions leaves them out).  Shading units (material "
                              "arms, background, new sample) are VALU lane-INSTRUCTIONS of the unit compiled alone "
                              "(roofline.op_model.file): they include integer RNG hashing, moves and selects and both "
                              "sides of divergent branches, so a shading kernel's frac is an instruction-issue share, not an "
                              "FP64 FLOP fraction, and is not comparable with the traversal kernel's",
                "kernels": kern,
                "useful_ops_per_launch": int(ops / launches), "useful_ops_per_ray": round(ops / max(cst["rays"], 1), 1),
                "rays_through_the_traversal_kernel": int(traversal_rays(cst)) if not st["local_pool"] else None,
                "records_per_ray": round(cst["interior_visits"] / max(cst["rays"], 1), 2),
                "prim_tests_per_ray": round(prims / max(cst["rays"], 1), 2),
                "lane_utilisation": util,
                "hot_group": None if not cst.get("hot_group") else {
                    "rays_that_owed_the_test_per_ray": round(cst["hot_lane"] / max(cst["rays"], 1), 3),
                    "primitive_tests_per_ray": round(cst["hot_prim_tests"] / max(cst["rays"], 1), 3),
                    "triangle_tests_that_went_on_to_divide_per_ray": round(cst["hot_tri_divided"] / max(cst["rays"], 1), 4),
                    "queries_answered_by_the_kernel_that_made_the_ray": round(cst["pre_rays"] / max(cst["rays"], 1), 3)},
                # SURVEY.md 8(d)'s figure: record fetches of an incoherent walk, served by LDS / L1 / L2 / Infinity
                # Cache -- it can exceed the HBM peak and bounds nothing; kept for comparison
                "algorithmic_bytes_per_launch": int(abytes / launches),
                "algorithmic_bytes_per_ray": round(abytes / max(cst["rays"], 1), 1),
                "algorithmic_GBps": round(abytes / (dom_ms * 1e-3) / 1e9, 1),
                # from the PMC passes of scripts/profile_round.sh, only when taken on this very source tree and
                # workload: bytes between L2 and the fabric (32 B x TCC_EA0_RDREQ_DRAM_32B + WRREQ_WRITE_DRAM_32B,
                # Infinity-Cache hits included; calibrated exact on 64/128/192-byte records, profiles/) and the
                # share of SIMD cycles with a vector instruction issued (SQ_INSTS_VALU x 4 cycles)
                "fabric_GBps": fabric_gbs,
                "fabric_frac_of_hbm_peak": None if fabric_gbs is None else round(fabric_gbs / HBM_PEAK_GBS, 4),
                "fp64_valu_busy": valu_busy, "valu_wave_instructions_per_ray": valu_per_ray,
                "valu_wave_instructions_per_traversal_ray": valu_per_trav_ray, "pmc_source": pmc_src,
            }
        return {"cam": cam, "W": W, "H": H, "chunk": chunk, "value": total_rays / max_elapsed / 1e6, "flights": F,
                "warm_frames": n_warm, "unpipelined_ms": None if unpipelined_s is None else unpipelined_s * 1e3,
                "frame_alone_ms": alone_ms,
                "ms_per_step": max_elapsed / steps * 1e3, "rays_per_step": int(total_rays / steps),
                "checksum": checksum, "sha": fb_sha, "roofline": roofline, "shares": shares,
                "exact_walk": int(st["exact_walk"]), "local_pool": int(st["local_pool"]), "hot_group": int(st["hot_group"])}

    main_run = measure(cx, cam_args, args.steps, args.warmup, not args.no_roofline, fast=args.fast_traversal)
    W, H, chunk = main_run["W"], main_run["H"], main_run["chunk"]

    secondary = None
    if (args.config == 5 and args.camera == "reference" and world == 1 and not reduced and not args.no_secondary
            and not args.no_roofline):
        close = scenes.camera_for_resolution(scenes.MESH_CLOSE_CAM, W0, H0)
        sec = measure(cx, close, 3, 1, True)
        secondary = {"workload": "the same scene and settings from a camera the mesh fills (scenes.MESH_CLOSE_CAM)",
                     "steps": 3, "warmup": 1,
                     "value": round(sec["value"], 2), "unit": "Mray/s", "ms_per_step": round(sec["ms_per_step"], 2),
                     "rays_per_step": sec["rays_per_step"], "ray_shares": sec["shares"],
                     "records_per_ray": sec["roofline"]["records_per_ray"],
                     "prim_tests_per_ray": sec["roofline"]["prim_tests_per_ray"],
                     "roofline_frac": sec["roofline"]["frac"]}

    fast_block = None
    if (args.config == 5 and args.camera == "reference" and world == 1 and not reduced and not args.no_fast
            and not args.no_roofline and not args.fast_traversal and not main_run["local_pool"]):
        fr = measure(cx, cam_args, 3, 1, True, fast=True)
        # (reported, not asserted: the headline line must not be lost to its companion measurement; the tests pin it)
        same = fr["sha"] == main_run["sha"] and fr["exact_walk"] == 0 and main_run["exact_walk"] == 1
        if not same:
            print("bench.py: the fast walk's frame or walk flags differ from the default walk's", file=sys.stderr)
        fast_block = {"workload": "the headline frame by rayrs_render_params.fast_traversal = 1: closest-hit culling and tight "
                                  "leaf boxes, two bets on the reference's arithmetic (measured, not proved: include/rayrs_hip.h); "
                                  "the headline itself is the default walk, the reference's visit set by construction",
                      "steps": 3, "warmup": 1, "value": round(fr["value"], 2), "unit": "Mray/s",
                      "ms_per_step": round(fr["ms_per_step"], 2), "rays_per_step": fr["rays_per_step"],
                      "records_per_ray": fr["roofline"]["records_per_ray"],
                      "prim_tests_per_ray": fr["roofline"]["prim_tests_per_ray"],
                      "traversal_ms": fr["roofline"]["kernels"]["wf_trav_kernel"]["ms"],
                      "roofline_frac": fr["roofline"]["frac"], "lane_utilisation": fr["roofline"]["lane_utilisation"],
                      "framebuffer_sha256": fr["sha"], "same_frame_as_headline": bool(same)}

    # BASELINE.json configs[0..3] at full size beside the headline (on the driver's record, not only in
    # profiles/): 1 untimed + 3 timed frames each, the kernel with the largest share of the step priced as the headline's is
    configs_block = None
    if (args.config == 5 and args.camera == "reference" and world == 1 and not reduced and not args.no_configs
            and not args.no_roofline and not args.fast_traversal):
        configs_block = []
        for n in (1, 2, 3, 4):
            c = make_context(n, 1)
            r = measure(c, c.cam_args, 3, 1, True)
            rf = r["roofline"]
            configs_block.append({
                "config": f"configs[{n - 1}]", "resolution": [r["W"], r["H"]], "spp": c.spp, "max_bounces": c.max_bounces,
                "primitives": c.info["n_prims"], "steps": 3, "warmup": 1,
                "value": round(r["value"], 2), "unit": "Mray/s", "ms_per_step": round(r["ms_per_step"], 3),
                "rays_per_step": r["rays_per_step"],
                "route": "local pool (one launch, paths resident in LDS)" if r["local_pool"] else
                         ("streaming, default walk" + (" + hot group" if r["hot_group"] else "")),
                "kernel": rf["kernel"], "kernel_share_of_step": rf["kernel_share_of_step"], "frac": rf["frac"],
                "frac_ref_flops": rf.get("frac_ref_flops"),
                "framebuffer_sha256": r["sha"], "scene_build_s": round(c.build_s, 2)})
            del c, r

    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import _oracle
        ncores = effective_cores()
        t0 = time.time()
        osc = _oracle.OracleScene(objs, 1e-6, 1e6, heur, hdri, builder=1)
        ocam = _oracle.OracleCamera(*cam_args)
        obuild = time.time() - t0
        mid = H // 2
        # calibrate on a thin band, then size the sample for ~cpu-seconds of wall time
        band = max(16, H // 64)
        _, cal = osc.render(ocam, 1, max_bounces, seed=0x5EED, rows=(mid - band // 2, mid + band // 2),
                            nthreads=ncores, traversal=0)
        rate = cal["rays"] / max(cal["seconds"], 1e-6)
        rays_per_row_spp = cal["rays"] / band
        rows = min(H, max(band, 16 * (int(H // 8) // 16)))
        want = rate * args.cpu_seconds
        cspp = int(max(1, min(spp, want / max(rays_per_row_spp * rows, 1.0))))
        _, cst2 = osc.render(ocam, cspp, max_bounces, seed=0x5EED, rows=(mid - rows // 2, mid + rows // 2),
                             nthreads=ncores, traversal=0)
        cpu_baseline = {"value": round(cst2["rays"] / cst2["seconds"] / 1e6, 4), "unit": "Mray/s", "cores": ncores,
                        "kind": "port",
                        "sample": f"C restatement of rayrs-lib's CPU path (recursive un-narrowed BVH traversal, "
                                  f"f64, 16x16 blocks on {ncores} threads), same scene/camera/seed, image rows "
                                  f"{mid - rows // 2}..{mid + rows // 2} of {H} at {cspp} spp: {cst2['rays']} rays in "
                                  f"{cst2['seconds']:.2f} s (oracle BVH build {obuild:.1f} s not timed)"}

    collective = None
    if use_dist:
        rccl = sorted({l.split()[-1] for l in open("/proc/self/maps") if "librccl" in l})
        collective = {"backend": args.backend, "world": world, "forced_single_rank": bool(args.force_dist and world == 1),
                      "librccl_mapped": rccl[0] if rccl else None}
    if rank == 0:
        n_prims = info["n_prims"]
        what = {1: "floor + 1 diffuse sphere", 2: "floor + 7 Cook-Torrance metallic spheres",
                3: f"floor + {n_prims - 2}-triangle PLY mesh + area light",
                4: "floor + 7 Cook-Torrance frosted-glass spheres",
                5: f"floor + {n_prims - 1}-triangle PLY mesh"}[args.config]
        line = {
            "metric": "Mray/s (primary+secondary) on 1M-tri scene @1024spp" if args.config == 5 else
                      f"Mray/s (primary+secondary), BASELINE.json configs[{args.config - 1}]",
            "value": round(main_run["value"], 2),
            "unit": "Mray/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(main_run["ms_per_step"], 2),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"configs[{args.config - 1}]: {what}, {W}x{H}, {spp} spp, max {max_bounces} bounces, "
                            f"SAH(1000) BVH, 1024x512 procedural HDRI, "
                            + ("the reference's obj_scene camera (test_scenes.rs:91-99)" if args.config in (3, 5)
                               and args.camera == "reference" else f"camera: {args.camera}")
                            + (" [REDUCED: development run]" if reduced else ""),
                "resolution": [W, H], "spp": spp, "max_bounces": max_bounces, "primitives": n_prims,
                "sample_chunk": chunk, "parallelism": f"8x8 image tiles interleaved over {world} GPU(s), "
                                                      f"scene replicated, one RCCL reduce of the f32x3 framebuffer",
                # a step is one whole frame; with more than one in flight consecutive steps overlap (the end of a frame
                # runs beside the start of the next), so ms_per_step = timed region / steps is a throughput figure and
                # frame_alone_ms the latency of a frame rendered with the GPU to itself
                "frames_in_flight": main_run["flights"],
                "untimed_frames_before_the_timed_region": main_run["warm_frames"],  # = warmup, or one per flight if that is more
                "frame_alone_ms": None if main_run["frame_alone_ms"] is None else round(main_run["frame_alone_ms"], 2),
                # the same work without the overlap: two more frames rendered one after the other on one flight, render +
                # framebuffer reduce each, max over ranks -- the figure to hold against a 1-GPU line's ms_per_step
                "unpipelined_ms_per_frame": None if main_run["unpipelined_ms"] is None else round(main_run["unpipelined_ms"], 2),
                # which walk answered the timed frames' BVH queries (rayrs_render_stats.exact_walk)
                "walk": ("local pool: the gate tree's groups, nothing culled (the reference's visit set)" if main_run["local_pool"] else
                         ("default: the gate tree, nothing culled -- the reference's visit set by construction"
                          + ("; root box, hot group (the floor's bottom Node) and the tree's first record are tested by the kernels that make the rays" if main_run["hot_group"] else ""))
                         if main_run["exact_walk"]
                         else "fast_traversal: closest-hit culling + tight leaf boxes (two bets)"),
                "layout": "compact f32 records" if info["compact"] else "f64 records",
                "bvh_depth": info["depth"], "walk_tree_records": info["n_wide"] if args.fast_traversal else (info["hot_n_wide"] if main_run["hot_group"] else info["gate_n_wide"]),
                "hbm_bytes_per_rank": cx.hbm_bytes,
                "scene_bytes": info["device_bytes"],
                "scene_build_s": round(build_s, 2), "source_hash": source_hash(),
                "workload_key": workload_key(cx, W, H, chunk), "collective": collective,
            },
            "rays_per_step": main_run["rays_per_step"],
            "framebuffer_checksum": main_run["checksum"],
            "framebuffer_sha256": main_run["sha"],
            # what the BVH queries of the frame found: the headline camera is the reference's own framing, from
            # which most queries end on the floor rectangle, not in the mesh -- see `secondary`
            "ray_shares": main_run["shares"],
            "roofline": main_run["roofline"],
            "cpu_baseline": cpu_baseline,
            "secondary": secondary,
            "fast": fast_block,
            "configs": configs_block,
        }
        print(json.dumps(line), flush=True)

    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
