"""The default walk's hot group and the pre-test of new rays (layout.h HotGroupDev, device_path.h hot_group_step,
wavefront.hip finish_rays) on the GPU.

The hot group -- on the obj scenes the floor's bottom Node: the 50 x 50 rectangle and the three mesh triangles the
reference's builder left beside it -- is tested once per ray, outside the tree, by the kernel that MAKES the ray, for a
whole batch at once, together with the first record of the tree without it; a ray that enters none of that record's slots
never travels through the traversal kernel.  What must hold: the frame, the ray counts and the primitive-test counts are
those of the walk over the whole gate tree (rayrs_lab hot_group = 0xffffffff) and of the oracle; rays aimed AT the group's
primitives, where the early rejection is not settled and the divisions are made, return the recursion's hits bit for bit;
and how the frame is cut up -- pool size, sample chunk, tile share -- changes nothing."""
import numpy as np
import pytest

import _oracle
import rayrs_amd
from rayrs_amd import _ffi, procedural, scenes

pytestmark = pytest.mark.gpu

HDRI = procedural.make_hdri(256, 128)


def bits(x):
    return np.ascontiguousarray(x).view(np.uint64)


def make(level=4, w=64, h=48, **kw):
    cam_args, objs, heur = scenes.mesh_scene(level, **kw)
    cam_args = scenes.camera_for_resolution(cam_args, w, h)
    scene = rayrs_amd.Scene(objs, 1e-6, 1e6, heur, HDRI, device=0)
    osc = _oracle.OracleScene(objs, 1e-6, 1e6, heur, HDRI)
    return scene, rayrs_amd.Camera(*cam_args), osc, _oracle.OracleCamera(*cam_args), objs


def gpu_intersect(scene, o, d, exact=1):
    o, d = np.ascontiguousarray(o, dtype=np.float64), np.ascontiguousarray(d, dtype=np.float64)
    t = np.zeros(len(o))
    obj = np.zeros(len(o), dtype=np.int64)
    _ffi.check(scene._L.rayrs_test_intersect(scene._h, o.ctypes.data, d.ctypes.data, len(o), int(exact), t.ctypes.data,
                                             obj.ctypes.data), "rayrs_test_intersect")
    return t, obj


def test_the_frame_and_the_tests_made_are_those_of_the_whole_gate_tree():
    scene, cam, osc, ocam, _ = make(4, 96, 64)
    assert scene.info()["hot_count"] >= 1
    img, st = rayrs_amd.render(scene, cam, 8, out_f64=True, count_work=True)
    assert st["hot_group"] == 1 and st["exact_walk"] == 1 and st["hot_lane"] > 0 and 0 < st["pre_rays"] < st["rays"]
    assert st["hot_prim_tests"] == scene.info()["hot_count"] * (st["hot_prim_tests"] // scene.info()["hot_count"]) > 0
    assert st["hot_tri_divided"] < st["hot_prim_tests"] // 4   # most of the group's triangle tests are settled before the divisions
    scene.lab_set(hot_group=0xffffffff)
    ref, rst = rayrs_amd.render(scene, cam, 8, out_f64=True, count_work=True)
    assert rst["hot_group"] == 0 and rst["hot_lane"] == 0 and rst["pre_rays"] == 0
    scene.lab_set()
    assert np.array_equal(bits(img), bits(ref))
    for k in ("rays", "paths", "tri_tests", "sphere_tests", "plane_tests", "escaped_paths"):
        assert st[k] == rst[k], k
    assert st["interior_visits"] < rst["interior_visits"]
    # every ray that enters the root box is put to the group's gate exactly once (bounced rays that leave the scene miss it)
    assert 0 < st["hot_lane"] <= st["rays"] - st["direct_rays"]
    # the oracle: its recursion (the parity claim) and its walk of the product's records with the group beside them (the counters)
    oref, ost = osc.render(ocam, 8, traversal=0)
    assert np.array_equal(bits(img), bits(oref)) and st["rays"] == ost["rays"]
    _, wst = osc.use_product_walk(scene).render(ocam, 8, traversal=2)
    for k in ("rays", "interior_visits", "tri_tests", "sphere_tests", "plane_tests", "escaped_paths"):
        assert st[k] == wst[k], k
    # the fast walk does not know the group
    _, fst = rayrs_amd.render(scene, cam, 8, out_f64=True, fast_traversal=True)
    assert fst["hot_group"] == 0


def test_rays_aimed_at_the_groups_primitives_return_the_recursions_hits():
    """Origins all around, directions through points on and just beside the hot group's primitives (within two of
    their sizes: barycentric coordinates inside [-2, 3], where the phase's early rejection is NOT settled and the three
    divisions are made), plus rays in the primitives' planes and along their edges."""
    scene, cam, osc, ocam, objs = make(4)
    info = scene.info()
    pp = scene.export_bvh()[2]
    boxes = osc.object_boxes(info["n_prims"])
    r = np.random.default_rng(5)
    os_, ds_ = [], []
    for k in range(info["hot_count"]):
        b = boxes[int(pp[info["hot_first"] + k])]
        lo, hi = b[0::2], b[1::2]
        ext = np.maximum(hi - lo, 1e-3)
        if ext.max() > 10.0:          # the floor: aim all over it and over its edges
            target = r.uniform(lo - 1.0, hi + 1.0, (6000, 3))
        else:
            target = r.uniform(lo - 2.0 * ext, hi + 2.0 * ext, (6000, 3))
            target[:1500] = r.uniform(lo, hi, (1500, 3))
        o = r.uniform(-8.0, 8.0, (6000, 3))
        o[:, 1] = np.abs(o[:, 1]) + 0.05
        o[::5] = target[::5] + r.standard_normal((1200, 3)) * ext.max() * 3.0   # from close by
        os_.append(o), ds_.append(target - o)
        # in the primitive's plane (grazing) and exactly axis-aligned through its box
        g = r.uniform(lo, hi, (500, 3))
        gd = r.uniform(lo, hi, (500, 3)) - g
        os_.append(g - 3.0 * gd), ds_.append(gd)
    o, d = np.concatenate(os_), np.concatenate(ds_)
    d[(d == 0).all(axis=1)] = (0.0, -1.0, 0.0)
    rt, robj = osc.intersect_many(o, d, 1e-6, 1e6, traversal=0)
    t, obj = gpu_intersect(scene, o, d)
    hot_objs = {int(pp[info["hot_first"] + k]) for k in range(info["hot_count"])}
    per = {h: int((robj == h).sum()) for h in hot_objs}
    assert all(v > 100 for v in per.values()), per      # every primitive of the group is hit by some of these rays
    assert np.array_equal(obj, robj)
    assert np.array_equal(bits(t), bits(rt))
    scene.lab_set(hot_group=0xffffffff)
    t2, obj2 = gpu_intersect(scene, o, d)
    assert np.array_equal(obj2, robj) and np.array_equal(bits(t2), bits(rt))


@pytest.mark.parametrize("how", [dict(pool_slots=1024), dict(sample_chunk=1), dict(sample_chunk=5, pool_slots=4096), dict(ranks=3),
                                 dict(lab=dict(refill_min=64, leaf_min=1)), dict(lab=dict(stack_lds=2))],
                         ids=["tiny_pool", "chunk_1", "chunk_5_small_pool", "three_tile_shares", "thresholds", "stack_in_hbm"])
def test_how_the_frame_is_cut_up_changes_nothing(how):
    """A tiny pool (many rounds: slots answered by the kernels that make the rays wait a round in state HIT or MISS), other
    sample chunks, tile shares, other thresholds: the same frame, the same counters."""
    scene, cam, osc, ocam, _ = make(3, 61, 43, area_light=True)
    spp = 7
    chunk = how.get("sample_chunk", 0)
    oref, ost = osc.render(ocam, spp, 50, sample_chunk=chunk, traversal=0)
    _, wst = osc.use_product_walk(scene).render(ocam, spp, 50, sample_chunk=chunk, traversal=2)
    scene.set_tuning(pool_slots=how.get("pool_slots", 0))
    scene.lab_set(**how.get("lab", {}))
    ranks = how.get("ranks", 1)
    img, tot = None, {}
    for r in range(ranks):
        img, st = rayrs_amd.render(scene, cam, spp, 50, sample_chunk=chunk, out_f64=True, count_work=True, tile_rank=r, tile_ranks=ranks, out=img)
        assert st["hot_group"] == 1
        for k in ("rays", "paths", "interior_visits", "tri_tests", "plane_tests", "escaped_paths", "pre_rays", "hot_lane"):
            tot[k] = tot.get(k, 0) + st[k]
    assert np.array_equal(bits(img), bits(oref))
    for k in ("rays", "interior_visits", "tri_tests", "plane_tests", "escaped_paths"):
        assert tot[k] == wst[k], k
    assert tot["rays"] == ost["rays"] and 0 < tot["pre_rays"] < tot["rays"]


@pytest.mark.parametrize("lab", [dict(leaf_min=64, leaf_wait=64), dict(leaf_min=64, leaf_wait=64, refill_min=64, stack_lds=2),
                                 dict(leaf_min=1, leaf_wait=1), dict(leaf_min=64, leaf_wait=1, refill_min=1)],
                         ids=["queues_fill", "queues_fill_stack_in_hbm", "leaf_phase_at_once", "one_lane_waits"])
def test_leaf_groups_set_aside_in_any_order_change_nothing(lab):
    """The default walk's lanes set the leaf groups they reach aside and walk on (device_path.h trav_interior_step_defer): the
    closest hit is the smallest accepted t, first primitive in depth-first order on ties, in ANY order of the groups.  With leaf
    phases held back until nobody has a record to visit the lanes' queues fill (a lane whose queue could not take four more
    groups sits out interior phases: the queue cannot overflow); with a leaf phase at every group nothing ever waits.  Same
    frame, same counters -- on a mesh whose rays cross a dozen groups."""
    scene, cam, osc, ocam, _ = make(5, 48, 36, area_light=True)
    spp = 5
    oref, ost = osc.render(ocam, spp, 50, traversal=0)
    _, wst = osc.use_product_walk(scene).render(ocam, spp, 50, traversal=2)
    scene.lab_set(**lab)
    img, st = rayrs_amd.render(scene, cam, spp, 50, out_f64=True, count_work=True)
    assert st["hot_group"] == 1
    assert np.array_equal(bits(img), bits(oref))
    for k in ("rays", "interior_visits", "tri_tests", "plane_tests", "escaped_paths"):
        assert st[k] == wst[k], k
    assert st["rays"] == ost["rays"]
    # the same on the close camera, where most rays need a deep walk
    cam2 = rayrs_amd.Camera(*scenes.camera_for_resolution(scenes.MESH_CLOSE_CAM, 40, 40))
    ocam2 = _oracle.OracleCamera(*scenes.camera_for_resolution(scenes.MESH_CLOSE_CAM, 40, 40))
    osc2 = osc.use_product_walk(scene)
    oref2, ost2 = osc2.render(ocam2, 3, 50, traversal=0)
    img2, st2 = rayrs_amd.render(scene, cam2, 3, 50, out_f64=True)
    assert np.array_equal(bits(img2), bits(oref2)) and st2["rays"] == ost2["rays"]


def test_a_group_of_spheres_and_a_rectangle_can_be_the_hot_group():
    """The phase tests whatever kinds the group holds: a floor that shares its bottom Node with spheres."""
    from rayrs_amd.api import Axis, BvhHeuristic, Emission, Material, Object
    r = np.random.default_rng(3)
    grey = Material.LambertianDiffuse((0.7, 0.7, 0.7))
    objs = [Object.plane(Axis.Y, -25.0, 25.0, -25.0, 25.0, 0.0, grey, Emission.Dark())]
    for i in range(60):
        c = r.uniform(-2.0, 2.0, 3)
        c[1] = abs(c[1]) + 0.3
        objs.append(Object.sphere(0.2, c, Material.Reflect((0.9, 0.9, 0.9)) if i % 2 else grey, Emission.Dark()))
    cam_args = scenes.camera_for_resolution(scenes.mesh_scene(2)[0], 64, 48)
    for heur in (BvhHeuristic.Sah(1000), BvhHeuristic.Midpoint):
        scene = rayrs_amd.Scene(objs, 1e-6, 1e6, heur, HDRI, device=0)
        info = scene.info()
        assert info["hot_count"] >= 1
        osc = _oracle.OracleScene(objs, 1e-6, 1e6, heur, HDRI)
        img, st = rayrs_amd.render(scene, rayrs_amd.Camera(*cam_args), 6, out_f64=True, count_work=True)
        ref, ost = osc.render(_oracle.OracleCamera(*cam_args), 6, traversal=0)
        assert st["hot_group"] == 1 and np.array_equal(bits(img), bits(ref)) and st["rays"] == ost["rays"]
        _, wst = osc.use_product_walk(scene).render(_oracle.OracleCamera(*cam_args), 6, traversal=2)
        for k in ("interior_visits", "tri_tests", "sphere_tests", "plane_tests"):
            assert st[k] == wst[k], k
