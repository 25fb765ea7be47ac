"""Stress of the parity claim: many random scenes (tests/test_gpu_render.py::_random_scene) rendered on the GPU
and by the oracle TWICE -- with the reference's recursion (traversal=0: BvhTree::intersect, bvh.rs:391-415, on the
oracle's own restatement of the reference's tree; frame bits, ray / path / escaped-path counts) and with the
kernel's walk on the product's exported records (traversal=2; the work counters too).  The first comparison is
the parity claim; the second only says the counters of the roofline are the walk's.  Every third scene is also
rendered on the other route (rayrs_tuning.local_pool toggled) when the scene qualifies.  GPU box only.
usage: python scripts/fuzz_parity.py [first_seed] [count]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import _oracle, rayrs_amd
from rayrs_amd import procedural, scenes
import test_gpu_render as T

first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
HDRI = procedural.make_hdri(256, 128)
bad = 0
n_both = 0
n_far = 0
t0 = time.time()
for seed in range(first, first + count):
    r = np.random.default_rng(seed ^ 0xABCDEF)
    w, h = int(r.integers(9, 90)), int(r.integers(9, 70))
    spp = int(r.integers(1, 13))
    chunk = int(r.choice([0, 1, 3, 4, 5]))
    mb = int(r.choice([1, 2, 5, 50]))
    cam_args, objs, heur = T._random_scene(seed)
    if seed % 4 == 2:  # a handful of primitives: at most one walk-tree record, rendered by the local pool
        objs = objs[:int(r.integers(2, 10))]
    if seed % 3 == 0:  # no emitters: the one-line slot
        from rayrs_amd.api import Emission, Object
        for o in objs:
            o.emission = Emission.Dark()
    scene = rayrs_amd.Scene(objs, 1e-6, 1e6, heur, HDRI, device=0)
    if seed % 7 == 3:  # the same view through a long lens from 9 .. 40 scene diagonals away: such a frame takes the exact walk
        box = np.array(scene.info()["root_box"])
        diag = float(np.linalg.norm(box[1::2] - box[0::2]))
        o, up, look, fov = np.array(cam_args[0], dtype=float), cam_args[1], np.array(cam_args[2], dtype=float), cam_args[3]
        back = o - look
        dist = float(np.linalg.norm(back))
        if np.isfinite(diag) and diag > 0 and dist > 0:
            new = float(r.uniform(9.0, 40.0)) * diag + dist
            cam_args = (tuple(look + back / dist * new), up, tuple(look), fov * dist / new) + tuple(cam_args[4:])
    cam_args = scenes.camera_for_resolution(cam_args, w, h)
    cam = rayrs_amd.Camera(*cam_args)
    osc = _oracle.OracleScene(objs, 1e-6, 1e6, heur, HDRI)
    ocam = _oracle.OracleCamera(*cam_args)
    if seed % 5 == 0:
        scene.set_tuning(pool_slots=int(r.integers(1, 40)) * 1024)
    fast = seed % 2 == 0   # every other scene by rayrs_render_params.fast_traversal, the others by the default walk
    img, st = rayrs_amd.render(scene, cam, spp, mb, seed=seed, sample_chunk=chunk, out_f64=True, count_work=True, fast_traversal=fast)
    ref0, ost0 = osc.render(ocam, spp, mb, seed=seed, sample_chunk=chunk, traversal=0)   # the reference's recursion
    if seed % 4 == 1:   # ... and the other walk as well: the same frame
        imgx, stx = rayrs_amd.render(scene, cam, spp, mb, seed=seed, sample_chunk=chunk, out_f64=True, fast_traversal=not fast)
        if not (np.array_equal(imgx.view(np.uint64), ref0.view(np.uint64)) and stx["rays"] == ost0["rays"]):
            bad += 1
            print("MISMATCH (the other walk) seed", seed, flush=True)
    ok = np.array_equal(img.view(np.uint64), ref0.view(np.uint64))
    for k in ("rays", "paths", "escaped_paths"):
        ok = ok and st[k] == ost0[k]
    # the oracle's walk on the records the frame's queries walked: the gate tree with nothing culled unless the frame
    # took the fast walk (asked for, the streaming route, the camera not far from the scene: abi.cpp camera_is_far)
    n_far += fast and bool(st["exact_walk"]) and not st["local_pool"]
    ref, ost = osc.use_product_walk(scene, fast=not st["exact_walk"]).render(ocam, spp, mb, seed=seed, sample_chunk=chunk, traversal=2)
    ok = ok and np.array_equal(img.view(np.uint64), ref.view(np.uint64))
    for k in ("rays", "paths", "escaped_paths", "interior_visits", "tri_tests", "sphere_tests", "plane_tests"):
        ok = ok and st[k] == ost[k]
    if seed % 3 == 1 and scene.info()["local_pool"]:  # the streaming kernels on a scene the local pool renders
        scene.set_tuning(local_pool=1)
        img2, st2 = rayrs_amd.render(scene, cam, spp, mb, seed=seed, sample_chunk=chunk, out_f64=True, count_work=True, fast_traversal=fast)
        ok = ok and np.array_equal(img2.view(np.uint64), ref0.view(np.uint64)) and st2["rays"] == ost0["rays"]
        n_both += 1
    if not ok:
        bad += 1
        print("MISMATCH seed", seed, w, h, spp, chunk, mb, {k: (st[k], ost[k]) for k in ("rays", "interior_visits")}, flush=True)
    if (seed - first) % 20 == 19:
        print(f"{seed - first + 1} scenes, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
print("done:", count, "scenes,", bad, "mismatches;", n_both, "scenes rendered on both routes;", n_far, "frames asked for the fast walk and took the default one for their camera's distance")
sys.exit(1 if bad else 0)
