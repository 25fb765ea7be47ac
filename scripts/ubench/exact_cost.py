"""The default walk's two bets priced on one box: the same frame by (a) the default walk -- single primitives behind their
own widened boxes, closest-hit culling; (b) round 3's walk -- the reference's groups behind their gating boxes, culling
(rayrs_lab.h gate_tree); (c) rayrs_render_params.exact_traversal -- the groups, nothing culled: the reference's visit set.
Work counters beside the times.  usage: python scripts/ubench/exact_cost.py <config> <res> <spp>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import rayrs_amd
from rayrs_amd import scenes, procedural
cfg, res, spp = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
cam_args, objs, heur, _, mb = scenes.config(cfg)
cam_args = scenes.camera_for_resolution(cam_args, res, res)
scene = rayrs_amd.Scene(objs, 1e-6, 1e6, heur, procedural.make_hdri(1024, 512), device=0)
cam = rayrs_amd.Camera(*cam_args)
chunk = rayrs_amd.frame_sample_chunk(res, res, spp)
rayrs_amd.render(scene, cam, 4, mb)
ref = None
WALKS = [("default", dict(), False), ("gate tree + culling", dict(gate_tree=1), False), ("exact_traversal", dict(), True)]
for name, lab, exact in WALKS + WALKS[::-1]:
    scene.lab_set(**lab)
    img, st = rayrs_amd.render(scene, cam, spp, mb, sample_chunk=chunk, exact_traversal=exact)
    _, cst = rayrs_amd.render(scene, cam, spp, mb, sample_chunk=chunk, exact_traversal=exact, count_work=True)
    if ref is None:
        ref = img.copy()
    prims = cst["tri_tests"] + cst["sphere_tests"] + cst["plane_tests"]
    print(f"{name:20s}: trace {st['trace_ms']:8.1f} ms  trav {st['kernel_ms']:8.1f}  hit {st['hit_ms']:6.1f} miss {st['miss_ms']:6.1f}  "
          f"Mray/s {st['rays'] / st['trace_ms'] / 1e3:7.1f}  records/ray {cst['interior_visits'] / cst['rays']:6.2f}  primitive tests/ray {prims / cst['rays']:6.2f}  "
          f"same_bits={bool((img.view('u4') == ref.view('u4')).all())}", flush=True)
