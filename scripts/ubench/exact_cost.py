"""The walks priced on one box: the same frame by (a) the default walk -- the reference's groups behind their gating boxes,
every member of an entered group tested, nothing culled: the reference's visit set by construction; (b) the fast walk
(rayrs_render_params.fast_traversal) -- single primitives behind their own clipped boxes, closest-hit culling: the two bets;
(c) rounds 2-3's walk -- the groups, culling (rayrs_lab.h gate_tree).  Work counters beside the times.
usage: python scripts/ubench/exact_cost.py <config> <res> <spp> [lab settings "k=v,k=v" applied to every walk]
ONLY=default,fast restricts the rows; CAMERA=close takes the mesh-filling camera."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import rayrs_amd
from rayrs_amd import scenes, procedural
cfg, res, spp = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
extra = {k: int(v, 0) for k, v in (kv.split("=") for kv in (sys.argv[4] if len(sys.argv) > 4 else "").split(",") if kv)}
cam_args, objs, heur, _, mb = scenes.config(cfg)
if os.environ.get("CAMERA") == "close":
    cam_args = scenes.MESH_CLOSE_CAM
cam_args = scenes.camera_for_resolution(cam_args, res, res)
scene = rayrs_amd.Scene(objs, 1e-6, 1e6, heur, procedural.make_hdri(1024, 512), device=0)
cam = rayrs_amd.Camera(*cam_args)
chunk = rayrs_amd.frame_sample_chunk(res, res, spp)
rayrs_amd.render(scene, cam, 4, mb)
ref = None
WALKS = [("default", dict(), False), ("whole gate tree", dict(hot_group=0xffffffff), False), ("fast", dict(), True),
         ("fast on the gate tree", dict(gate_tree=1), True)]
if os.environ.get("ONLY"):
    WALKS = [w for w in WALKS if w[0] in os.environ["ONLY"].split(",")]
for name, lab, fast in WALKS + WALKS[::-1]:
    scene.lab_set(**dict(extra, **lab))
    img, st = rayrs_amd.render(scene, cam, spp, mb, sample_chunk=chunk, fast_traversal=fast)
    _, cst = rayrs_amd.render(scene, cam, spp, mb, sample_chunk=chunk, fast_traversal=fast, count_work=True)
    if ref is None:
        ref = img.copy()
    prims = cst["tri_tests"] + cst["sphere_tests"] + cst["plane_tests"]
    print(f"{name:26s}: trace {st['trace_ms']:8.1f} ms  trav {st['kernel_ms']:8.1f}  hit {st['hit_ms']:6.1f} miss {st['miss_ms']:6.1f}  "
          f"Mray/s {st['rays'] / st['trace_ms'] / 1e3:7.1f}  records/ray {cst['interior_visits'] / cst['rays']:6.2f}  primitive tests/ray {prims / cst['rays']:6.2f}  "
          f"lanes {cst['step_lane'] / max(cst['step_wave'], 1):.2f}/{cst['inner_wave'] / max(cst['leaf_wave'], 1):.2f}  exact_walk={st['exact_walk']} hot_group={st['hot_group']}  "
          f"same_bits={bool((img.view('u4') == ref.view('u4')).all())}", flush=True)
