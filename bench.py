#!/usr/bin/env python
"""bench.py -- Mray/s of the MI355X radiance integrator on BASELINE.json's headline
configuration (configs[4]: floor + 1,310,720-triangle mesh, 2048x2048, 1024 spp,
50 bounces), on N GPUs of one node.

One step = one full frame: every 8x8 image tile of this rank pushed through the
gfx950 path pipeline (persistent traversal kernel + hit/miss kernels, one round
per bounce), the per-pixel resolve, and (N > 1) one RCCL reduce of the f32x3
framebuffer to rank 0.  Tiles are interleaved over ranks (tile t ->
rank t % N), the scene is replicated, total work is fixed: strong scaling.
`value` = BVH queries of all ranks / max-over-ranks wall time (scene build,
upload and file I/O excluded -- the region the reference times, main.rs:59-100).

Prints ONE JSON line on rank 0 (contract in the task statement) carrying
  roofline     -- algorithmic bytes of the traversal kernel per launch (from the
                  kernel's own counters on the flattened layout) / its average
                  HIP-event duration, against the 8 TB/s HBM peak;
  cpu_baseline -- the CPU oracle in reference mode (recursive un-narrowed
                  traversal over a pointer tree, rayrs-lib's algorithm) timed on
                  this box's host cores on a bounded band of the same frame.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# RCCL shares buffers between the ranks' processes through dmabuf IPC on this pool's driver
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec


def algorithmic_bytes(stats, info):
    """bytes the traversal kernel must move on the flattened layout:
    one record per interior visit and per primitive test, plus, per BVH query, the
    ray it reads from the path pool (origin + direction, 48 B), the result it
    writes back (t + primitive, 12 B) and the slot's state byte (read + write)."""
    return (stats["interior_visits"] * info["node_bytes"]
            + (stats["tri_tests"] + stats["sphere_tests"] + stats["plane_tests"]) * info["prim_bytes"]
            + stats["rays"] * (48 + 12 + 2))


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=5, help="BASELINE.json configs[n-1]; 5 is the headline")
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel (development only)")
    ap.add_argument("--res", type=int, default=0, help="override resolution (development only)")
    ap.add_argument("--sample-chunk", type=int, default=4,
                    help="smallest sample chunk to use (the library doubles it until the whole frame has at most "
                         "2^30 items); small chunks keep the end of a frame, and of a tile share, short")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend; 'gloo' (host reduce) is for "
                    "rehearsing the N>1 path on a box with fewer GPUs than ranks")
    ap.add_argument("--device", type=int, default=-1, help="force the HIP device of every rank (rehearsal only)")
    ap.add_argument("--no-build", action="store_true",
                    help="do not run make: required under rocprofv3 (a profiled process must not spawn the compiler); "
                         "fails if the library is older than its sources")
    args = ap.parse_args()

    import numpy as np
    import torch

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

    use_dist = world > 1
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

    # the mesh takes the route a scanned model would: written as a binary PLY, read back by the library's loader
    import tempfile
    ply_path = os.path.join(tempfile.gettempdir(), f"rayrs_bench_mesh_rank{rank}.ply") if args.config in (3, 5) else None
    cam_args, objs, heur, spp, max_bounces = scenes.config(args.config, ply_path=ply_path)
    reduced = False
    if args.spp:
        spp, reduced = args.spp, True
    if args.res:
        cam_args, reduced = scenes.camera_for_resolution(cam_args, args.res, args.res), True
    hdri = procedural.make_hdri(1024, 512)
    t0 = time.time()
    scene = rayrs_amd.Scene(objs, 1e-6, 1e6, heur, hdri, device=local_rank)
    build_s = time.time() - t0
    cam = rayrs_amd.Camera(*cam_args)
    info = scene.info()
    H, W = cam.y_pixels(), cam.x_pixels()

    # a pixel's samples are summed in chunks; the chunk comes from the WHOLE frame (rayrs_frame_sample_chunk:
    # at most 2^30 (pixel, chunk) items), never from the rank count, so every N renders the same bits
    args.sample_chunk = rayrs_amd.frame_sample_chunk(W, H, spp, args.sample_chunk)

    dev = torch.device("cuda", local_rank)
    fb = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev)
    params = rayrs_amd.make_params(spp, max_bounces, seed=0x5EED, sample_chunk=args.sample_chunk, tile_rank=rank,
                                   tile_ranks=world)

    def step():
        fb.zero_()
        rayrs_amd.render_launch(scene, cam, params, fb.data_ptr(), stream.cuda_stream)
        if use_dist and args.backend == "nccl":
            tiles.reduce_framebuffer(fb, dst=0)       # RCCL over xGMI, ordered after the render on this stream
        st = rayrs_amd.render_finish(scene)
        if use_dist and args.backend != "nccl":       # rehearsal: reduce through host memory
            host = fb.cpu()
            tiles.reduce_framebuffer(host, dst=0)
            fb.copy_(host)
        return st

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    t_begin = time.perf_counter()
    rays = 0
    kernel_ms = []
    for _ in range(args.steps):
        st = step()
        rays += st["rays"]
        kernel_ms.append(st["kernel_ms"])
    fence()
    elapsed = time.perf_counter() - t_begin

    tot = torch.tensor([float(rays), elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
    if use_dist:
        r = tot[:1].clone()
        e = tot[1:].clone()
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
        dist.all_reduce(e, op=dist.ReduceOp.MAX)
        total_rays, max_elapsed = float(r.item()), float(e.item())
    else:
        total_rays, max_elapsed = float(rays), elapsed
    checksum, fb_sha = 0.0, None
    if rank == 0:  # the assembled frame: identical bits for every rank count (tests/test_gpu_multi_process.py)
        import hashlib
        checksum = float(fb.double().sum().item())
        fb_sha = hashlib.sha256(fb.cpu().numpy().tobytes()).hexdigest()

    roofline = None
    if not args.no_roofline:
        # same launch once more with the traversal counters compiled in (untimed):
        # the work of a launch is a pure function of (scene, seed), so the counts
        # apply to the timed launches exactly
        pc = rayrs_amd.make_params(spp, max_bounces, seed=0x5EED, sample_chunk=args.sample_chunk, tile_rank=rank,
                                   tile_ranks=world, count_work=True)
        fb2 = torch.zeros_like(fb)
        rayrs_amd.render_launch(scene, cam, pc, fb2.data_ptr(), stream.cuda_stream)
        cst = rayrs_amd.render_finish(scene)
        assert cst["rays"] == st["rays"], "counting launch traced a different frame"
        # the dominant kernel is the traversal kernel, launched once per path round
        launches = st["kernel_launches"]
        # HBM bytes per launch come from PMC passes of this same command (they cannot be collected
        # in-process): profiles/r01_final_hbm_traffic.json, valid for the unreduced headline workload only
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "r01_final_hbm_traffic.json")
        if os.path.exists(tpath) and not reduced and args.config == 5 and world == 1:
            tj = json.load(open(tpath))
            k = tj["kernels"]["wf_trav_kernel"]
            if k["launches"] == launches and tj.get("sample_chunk") == args.sample_chunk:
                traffic = k["bytes_per_launch"]
                traffic_src = "profiles/r01_final_hbm_traffic.json (rocprofv3 FETCH_SIZE + WRITE_SIZE, calibrated)"
        # FP64-VALU occupancy of the same kernel, from a PMC pass of the same command (scripts/valu_pass.sh)
        valu_busy = None
        vpath = os.path.join(ROOT, "profiles", "r01_final_valu.json")
        if traffic is not None and os.path.exists(vpath):
            vk = json.load(open(vpath))["kernels"]["wf_trav_kernel"]
            if vk["launches"] == launches:
                valu_busy = vk["valu_busy"]
        abytes = algorithmic_bytes(cst, info) / launches          # per launch
        avg_ms = sum(kernel_ms) / len(kernel_ms) / launches       # per launch, HIP events on the render stream
        achieved = abytes / (avg_ms * 1e-3) / 1e9
        roofline = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
                    "kernel": "wf_trav_kernel", "launches_per_step": int(launches), "kernel_ms": round(avg_ms, 4),
                    "kernel_share_of_step": round(sum(kernel_ms) / len(kernel_ms) / (max_elapsed / args.steps * 1e3), 3),
                    "algorithmic_bytes_per_launch": int(abytes),
                    "bytes_per_ray": round(abytes * launches / max(cst["rays"], 1), 1),
                    "interior_visits_per_ray": round(cst["interior_visits"] / max(cst["rays"], 1), 2),
                    "prim_tests_per_ray": round((cst["tri_tests"] + cst["sphere_tests"] + cst["plane_tests"])
                                                / max(cst["rays"], 1), 2),
                    # The algorithmic bytes are record fetches of an incoherent tree walk over a 97 MB scene:
                    # LDS, L1, L2 and the Infinity Cache serve nearly all of them, which is how `achieved`
                    # can pass the HBM peak.  What reaches HBM is `traffic`; what the kernel actually
                    # runs out of is FP64 VALU issue (SURVEY.md 8(d)'s secondary ceiling).
                    "hbm_achieved": None if traffic is None else round(traffic / (avg_ms * 1e-3) / 1e9, 2),
                    "hbm_frac": None if traffic is None else round(traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                    "fp64_valu_busy": valu_busy,
                    "fp64_valu_source": None if valu_busy is None else
                    "profiles/r01_final_valu.json (rocprofv3 SQ_INSTS_VALU x 4 cycles / SIMD-cycles of the kernel)"}

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

    if rank == 0:
        value = total_rays / max_elapsed / 1e6
        n_tri = info["n_prims"]
        line = {
            "metric": "Mray/s (primary+secondary) on 1M-tri scene @1024spp",
            "value": round(value, 2),
            "unit": "Mray/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(max_elapsed / args.steps * 1e3, 2),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"configs[{args.config - 1}]: floor + {n_tri - 1 if args.config == 5 else n_tri}-primitive "
                            f"procedural mesh scene, {W}x{H}, {spp} spp, max {max_bounces} bounces, SAH(1000) BVH, "
                            f"1024x512 procedural HDRI" + (" [REDUCED: development run]" if reduced else ""),
                "resolution": [W, H], "spp": spp, "max_bounces": max_bounces, "primitives": n_tri,
                "sample_chunk": args.sample_chunk, "parallelism": f"8x8 image tiles interleaved over {world} GPU(s), "
                                                                 f"scene replicated, one RCCL reduce of the f32x3 framebuffer",
                "layout": "compact f32 records" if info["compact"] else "f64 records",
                "bvh_depth": info["depth"], "scene_bytes": info["device_bytes"], "scene_build_s": round(build_s, 2),
            },
            "rays_per_step": int(total_rays / args.steps),
            "framebuffer_checksum": checksum,
            "framebuffer_sha256": fb_sha,
            "roofline": roofline,
            "cpu_baseline": cpu_baseline,
        }
        print(json.dumps(line), flush=True)

    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
