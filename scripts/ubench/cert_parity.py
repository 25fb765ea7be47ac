"""Quick GPU check of the certified walk against the oracle: frames against the reference's recursion (traversal 0), work
counters against the oracle's certified walk on the product's exported records, intersections on the fuzz families."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import numpy as np
import rayrs_amd, _oracle
from rayrs_amd import scenes, procedural, _ffi
hdri = procedural.make_hdri(64, 32)
for level in (3, 5):
    cam_args, objs, heur = scenes.mesh_scene(level, area_light=True)
    cam_args = scenes.camera_for_resolution(cam_args, 96, 64)
    scene = rayrs_amd.Scene(objs, 1e-6, 1e6, heur, hdri, device=0)
    cam = rayrs_amd.Camera(*cam_args)
    osc = _oracle.OracleScene(objs, 1e-6, 1e6, heur, hdri)
    ocam = _oracle.OracleCamera(*cam_args)
    ref, rs = osc.render(ocam, 8, 50, traversal=0)
    for walk in ("certified", "reference", "fast"):
        img, st = rayrs_amd.render(scene, cam, 8, 50, out_f64=True, walk=walk, count_work=True)
        if walk == "certified":
            _oracle.set_cull_margin(float("inf"))
            o2, os2 = osc.use_cert_tree(scene).render(ocam, 8, 50, traversal=2)
            _oracle.set_cull_margin(2.0 ** -10)
        elif walk == "reference":
            _oracle.set_cull_margin(float("inf"))
            o2, os2 = osc.use_walk_tree(scene, gate=True).render(ocam, 8, 50, traversal=2)
            _oracle.set_cull_margin(2.0 ** -10)
        else:
            o2, os2 = osc.use_walk_tree(scene).render(ocam, 8, 50, traversal=2)
        keys = ("rays", "interior_visits", "tri_tests", "plane_tests", "sphere_tests")
        print(level, walk, "filtered", scene.info()["n_filtered"], "frame==recursion", np.array_equal(img.view(np.uint64), ref.view(np.uint64)),
              "counters", [(k, st[k], os2[k]) for k in keys if st[k] != os2[k]] or "equal", "walk", st["walk"], flush=True)
import fuzz_traversal as F
z = np.zeros((2, 2, 3), dtype=np.float32)
for seed in (1, 2, 3, 5, 79):
    objs, heur, scale, verts, idx = F.scene_for(seed)
    t0, t1 = 1e-6 * scale, 1e9 * scale
    scene = rayrs_amd.Scene(objs, t0, t1, heur, z, device=0)
    osc = _oracle.OracleScene(objs, t0, t1, heur, z)
    rr = np.random.default_rng(seed * 104729 + 5)
    og, dg = F.rays_for(rr, verts, scale, 20000)
    oz, dz, eps = F.grazing_rays(rr, verts, idx, scale, 60000)
    o, d = np.ascontiguousarray(np.vstack([og, oz])), np.ascontiguousarray(np.vstack([dg, dz]))
    rt, robj = osc.intersect_batch(o, d, t0, t1, traversal=0)
    for walk in (0, 1, 2):
        t = np.zeros(len(o)); obj = np.zeros(len(o), dtype=np.int64)
        _ffi.check(scene._L.rayrs_test_intersect(scene._h, o.ctypes.data, d.ctypes.data, len(o), walk, t.ctypes.data, obj.ctypes.data), "x")
        bad = int(((obj != robj) | (t.view(np.uint64) != rt.view(np.uint64))).sum())
        print("seed", seed, "walk", walk, "mismatches", bad, "of", len(o), "hits", int((robj >= 0).sum()), flush=True)
