"""The trees the kernels walk (rayrs_amd/csrc/scene_host.cpp build_walk_trees) return the
reference's hits.  CPU only: the product builds the trees host-side (device = -1), the oracle walks
them with the kernel's rules (traversal 2: the fast walk's tree nearest slot first, boxes beyond the closest hit
culled; the default walk's gate tree with nothing culled)
and is compared with its restatement of the reference's recursion (traversal 0, bvh.rs:391-415),
bit for bit, on the cases where a different topology could show: degenerate rays, unhittable flat
boxes, direct leaves, coincident and abutting primitives (equal-t ties, bvh.rs:62)."""
import numpy as np
import pytest

import _oracle
import rayrs_amd
from rayrs_amd import procedural, scenes
from rayrs_amd.api import Axis, BvhHeuristic, Emission, Material, Object

HDRI = procedural.make_hdri(32, 16)
NR, DARK = Material.NoReflect(), Emission.Dark()


def both_walks(objs, heur, o, d, t0=1e-6, t1=1e6):
    prod = rayrs_amd.Scene(objs, t0, t1, heur, HDRI, device=-1)
    osc = _oracle.OracleScene(objs, t0, t1, heur, HDRI).use_walk_tree(prod)
    rt, robj = osc.intersect_many(o, d, t0, t1, traversal=0)
    wt, wobj = osc.intersect_many(o, d, t0, t1, traversal=2)
    assert np.array_equal(wobj, robj)
    assert np.array_equal(wt.view(np.uint64), rt.view(np.uint64))
    # the default walk: the gate tree, nothing culled
    osg = _oracle.OracleScene(objs, t0, t1, heur, HDRI).use_walk_tree(prod, gate=True)
    try:
        _oracle.set_cull_margin(float("inf"))
        xt, xobj = osg.intersect_many(o, d, t0, t1, traversal=2)
    finally:
        _oracle.set_cull_margin(2.0 ** -10)
    assert np.array_equal(xobj, robj)
    assert np.array_equal(xt.view(np.uint64), rt.view(np.uint64))
    # ... and on a scene with a hot group, the tree without it with the group tested beside the walk
    if prod.info()["hot_count"]:
        osh = _oracle.OracleScene(objs, t0, t1, heur, HDRI).use_product_walk(prod, hot=True)
        try:
            _oracle.set_cull_margin(float("inf"))
            ht, hobj = osh.intersect_many(o, d, t0, t1, traversal=2)
        finally:
            _oracle.set_cull_margin(2.0 ** -10)
        assert np.array_equal(hobj, robj)
        assert np.array_equal(ht.view(np.uint64), rt.view(np.uint64))
    return robj


def random_rays(n, seed, spread=6.0):
    r = np.random.default_rng(seed)
    o = r.uniform(-spread, spread, (n, 3))
    d = r.normal(size=(n, 3))
    return o, d


def test_box_geom_coincident_faces():
    """Object::box_geom puts BOTH Y faces at lower_left.y (lib.rs:486-505): two rectangles in the same
    plane, every ray through them gets two equal t.  The first in depth-first order must win."""
    objs = Object.box_geom((-1.0, 0.5, -1.0), (1.0, 2.0, 1.0), NR, DARK)
    objs += Object.box_geom((1.0, 0.5, -1.0), (3.0, 2.0, 1.0), NR, DARK)  # abutting box: shared X face
    objs.append(Object.plane(Axis.Y, -25.0, 25.0, -25.0, 25.0, 0.5, NR, DARK))  # floor in the same plane again
    o, d = random_rays(3000, 5, 4.0)
    o[:1000, 1] = np.abs(o[:1000, 1]) + 2.5
    d[:1000] = (0.0, -1.0, 0.0)
    d[:500, 0] = np.random.default_rng(6).uniform(-0.3, 0.3, 500)
    for heur in (BvhHeuristic.Sah(1000), BvhHeuristic.Sah(3), BvhHeuristic.Midpoint):
        hit = both_walks(objs, heur, o, d)
        assert (hit >= 0).sum() > 150


def test_mesh_resting_on_the_floor_and_duplicated():
    """Triangles lying in the floor's plane, and the same mesh twice (every hit is an exact tie)."""
    verts, idx = procedural.blob_mesh(2)
    v = verts.astype(np.float64)
    v[:, 1] = np.maximum(v[:, 1] - v[:, 1].min(), 0.0)  # lowest vertices in the plane y = 0
    flat = v.copy()
    flat[:, 1] = 0.0                                    # a copy squashed into the floor's plane
    objs = [Object.plane(Axis.Y, -25.0, 25.0, -25.0, 25.0, 0.0, NR, DARK)]
    objs += Object.from_triangles(v, idx, NR, DARK)
    objs += Object.from_triangles(v, idx, NR, DARK)
    objs += Object.from_triangles(flat, idx, NR, DARK)
    o, d = random_rays(3000, 9, 3.0)
    o[:, 1] = np.abs(o[:, 1]) + 0.1
    d[:, 1] = -np.abs(d[:, 1])
    for heur in (BvhHeuristic.Sah(1000), BvhHeuristic.Midpoint):
        hit = both_walks(objs, heur, o, d)
        assert (hit >= 0).sum() > 300


def test_unhittable_flat_boxes_stay_unhittable():
    """A bottom Node of coplanar axis-aligned rectangles has a zero-thickness box, which the slab test
    never passes (tmax <= tmin, geometry.rs:458-513): the reference cannot hit those rectangles, and
    neither may the walk tree, whose interior boxes around them are not flat."""
    objs = [Object.plane(Axis.Y, float(i), float(i) + 0.9, 0.0, 1.0, 0.0, NR, DARK) for i in range(4)]
    objs += [Object.sphere(0.4, (float(i), 2.0, 0.5), NR, DARK) for i in range(6)]
    o = np.array([[0.5 + i * 0.01, 5.0, 0.5] for i in range(300)])
    d = np.tile(np.array([0.0, -1.0, 0.0]), (300, 1))
    for heur in (BvhHeuristic.Sah(1000), BvhHeuristic.Midpoint):
        prod = rayrs_amd.Scene(objs, 1e-6, 1e6, heur, HDRI, device=-1)
        osc = _oracle.OracleScene(objs, 1e-6, 1e6, heur, HDRI).use_walk_tree(prod)
        rt, robj = osc.intersect_many(o, d, 1e-6, 1e6, traversal=0)
        wt, wobj = osc.intersect_many(o, d, 1e-6, 1e6, traversal=2)
        assert np.array_equal(wobj, robj) and np.array_equal(wt.view(np.uint64), rt.view(np.uint64))


@pytest.mark.parametrize("level", [3, 5])
def test_mesh_scene_frames_and_counters(level):
    """Whole frames through both walks; the walk tree must also be the cheaper one."""
    cam_args, objs, heur = scenes.mesh_scene(level, area_light=True)
    cam_args = scenes.camera_for_resolution(cam_args, 48, 32)
    prod = rayrs_amd.Scene(objs, 1e-6, 1e6, heur, HDRI, device=-1)
    osc = _oracle.OracleScene(objs, 1e-6, 1e6, heur, HDRI).use_walk_tree(prod)
    ocam = _oracle.OracleCamera(*cam_args)
    ref, rs = osc.render(ocam, 4, 50, traversal=0)
    bin_, bs = osc.render(ocam, 4, 50, traversal=1)   # ordered walk of the reference's own two-child records
    walk, ws = osc.render(ocam, 4, 50, traversal=2)
    assert np.array_equal(ref.view(np.uint64), walk.view(np.uint64)) and rs["rays"] == ws["rays"]
    assert np.array_equal(ref.view(np.uint64), bin_.view(np.uint64))
    assert ws["interior_visits"] * 2 < bs["interior_visits"]  # four-slot records of a better tree: fewer visits


def test_deep_reference_tree_gives_a_shallow_walk_tree():
    """250 levels of two spheres each in the reference's tree (tests/test_gpu_render.py's chain);
    the walk tree over the same groups is built by surface area, not by the reference's splits."""
    objs = []
    for k in range(250):
        x = 1.5 ** k
        for j in range(2):
            objs.append(Object.sphere(0.25 * x if k > 3 else 0.2, (x * (1 + 0.01 * j), 1.0, 0.0), NR, DARK))
    prod = rayrs_amd.Scene(objs, 1e-6, 1e60, BvhHeuristic.Midpoint, HDRI, device=-1)
    info = prod.info()
    assert info["depth"] > 100
    r = np.random.default_rng(3)
    o = np.tile(np.array([-3.0, 1.5, 4.0]), (500, 1))
    d = np.stack([np.abs(r.normal(size=500)) + 0.2, r.normal(size=500) * 0.05, -np.abs(r.normal(size=500)) * 0.3], 1)
    hit = both_walks(objs, BvhHeuristic.Midpoint, o, d, 1e-6, 1e60)
    assert (hit >= 0).sum() > 100


def test_hostile_rays_and_scene_scales():
    """Axes with d == 0, -0, denormal-small or huge components, origins far outside the scene and exactly on
    round coordinates, at scene scales of 1e-3, 1 and 1e6: the walk tree must return the reference's hits."""
    r = np.random.default_rng(21)
    for scale in (1e-3, 1.0, 1e6):
        objs = []
        for i in range(300):
            c = r.uniform(-5, 5, 3) * scale
            if i % 2:
                objs.append(Object.sphere(float(r.uniform(0.1, 0.8)) * scale, c, NR, DARK))
            else:
                objs.append(Object.triangle(c, c + r.uniform(-1, 1, 3) * scale, c + r.uniform(-1, 1, 3) * scale, NR, DARK))
        n = 3000
        o = r.uniform(-8, 8, (n, 3)) * scale
        o[:300] *= 1e6
        o[300:600] = np.round(o[300:600] / scale) * scale
        d = r.normal(size=(n, 3))
        d[::5, 0] = 0.0
        d[1::5, 1] = -0.0
        d[2::7, 2] = 1e-300
        d[3::11, 0] = 1e-40
        d[4::13] *= 1e200
        d[(d == 0).all(axis=1)] = (0.0, 1.0, 0.0)
        d[:300] = -o[:300] + r.normal(size=(300, 3)) * scale
        hit = both_walks(objs, BvhHeuristic.Sah(1000), np.ascontiguousarray(o), np.ascontiguousarray(d), 1e-6 * scale,
                         1e12 * scale)
        assert (hit >= 0).sum() > 50, scale


def test_both_bets_on_slivers_flat_sheets_and_grazing_rays():
    """The fast walk's two bets (rayrs_render_params.fast_traversal: closest-hit culling at 1 + 2^-10, single primitives
    behind boxes widened by 1/64; the reference does neither, and neither does the default walk) against the reference's recursion where
    Moeller-Trumbore is least accurate: sliver triangles, nearly flat sheets, origins up to 1e6 scene sizes away,
    general directions down to 1e-7 rad over the sheet and rays aimed along a triangle's own plane.
    The fast walk must match on the general family and on grazing rays 1e-7 rad and more off the plane from
    within 8 root-box diagonals and 4096 small-primitive sizes of the scene (from farther out a frame takes the default
    walk whatever it asks for: abi.cpp camera_is_far) -- on the sheets of 6 ... 40 quads per side on which that was
    measured in round 4, with a floor under them, and on a sheet of 350 well-shaped quads per side; on a sheet of 320
    SLIVER quads per side (seed 11, round 5) its culling loses hits to in-plane rays at 1e-7 rad and more from ANY
    distance (a few in 10^4): counted here, not required -- which is why the fast walk is the caller's choice and not
    the default.  The default walk (gate tree, nothing culled) matches on every ray of every family, by construction.
    scripts/fuzz_traversal.py is the same over 1e8 rays; this is 1.6 M.  The probe measures the cull margin itself:
    how far in front of a box around it a hit's t can lie."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import fuzz_traversal as F
    hdri = np.zeros((2, 2, 3), dtype=np.float32)
    worst_general, hits, elsewhere, fine_slivers = -1.0, 0, 0, 0
    # seed 4: an exactly flat sheet, whose boxes nothing can enter (geometry.rs:474); 11 and 22: sheets of 250+ quads per side (a
    # camera 8 diagonals out stands thousands of primitive sizes away), 11 of slivers; 13: a floor fifty sheets wide under the sheet
    for seed in list(range(1, 9)) + [11, 13, 22]:
        objs, heur, scale, verts, idx = F.scene_for(seed)
        t0, t1 = 1e-6 * scale, 1e9 * scale
        prod = rayrs_amd.Scene(objs, t0, t1, heur, hdri, device=-1)
        osc = _oracle.OracleScene(objs, t0, t1, heur, hdri).use_walk_tree(prod)
        osg = _oracle.OracleScene(objs, t0, t1, heur, hdri).use_walk_tree(prod, gate=True)
        for name, o, d in F.families(seed, verts, idx, scale, 200_000 if seed < 10 else 60_000, prod.info()["root_box"],
                                     F.small_extent(verts, idx)):
            rt, robj = osc.intersect_batch(o, d, t0, t1, traversal=0)
            wt, wobj = osc.intersect_batch(o, d, t0, t1, traversal=2)
            same = (wobj == robj) & (wt.view(np.uint64) == rt.view(np.uint64))
            if F.required(F.fine_slivers(verts, idx), name):
                assert same.all(), (seed, name)
                hits += int((robj >= 0).sum())
            elif name in F.REQUIRED:
                fine_slivers += int((~same).sum())
            else:
                elsewhere += int((~same).sum())
            try:
                _oracle.set_cull_margin(float("inf"))
                xt, xobj = osg.intersect_batch(o, d, t0, t1, traversal=2)
            finally:
                _oracle.set_cull_margin(2.0 ** -10)
            assert np.array_equal(xobj, robj) and np.array_equal(xt.view(np.uint64), rt.view(np.uint64)), (seed, name)
            if name == "general":
                w, in_front, beyond = osc.cull_margin_probe(o, d, t0, t1)
                worst_general = max(worst_general, w)
                assert beyond == 0
    assert hits > 500_000
    assert 0 < elsewhere < 2000  # the bets do fail out there (a few in 1e4 rays aimed along a plane from far away)
    assert fine_slivers < 100  # ... and, on finely tessellated slivers (seed 11: fuzz_traversal.fine_slivers), from nearby too -- counted, bounded
    assert worst_general < 2.0 ** -40  # general rays: a hit precedes a box of its own by ulps only


def test_the_bets_are_heuristics_and_this_is_where_they_end():
    """The failures the fast walk (rayrs_render_params.fast_traversal) cannot exclude, pinned (found by
    scripts/fuzz_traversal.py), one of each kind; the default walk -- the gate tree, nothing culled -- returns the
    reference's answer.
    Culling: a ray within 1e-9 rad of a triangle's plane from 4600 triangle sizes away puts that triangle's t 2 % in
    front of its gating box (seed 79); a walk that has already found the neighbour behind it skips the box.
    Leaf boxes: a ray aimed along a triangle's plane from 130 000 scene sizes away, 1e-5 rad off it (seed 2): the
    reference's own Moeller-Trumbore accepts a hit on a neighbouring triangle that the ray passes beside by more than
    1/64 of its size -- 1.5e-8 of t in front of the triangle the ray does cross."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import fuzz_traversal as F
    hdri = np.zeros((2, 2, 3), dtype=np.float32)

    def walks(seed, o, d):
        objs, heur, scale, verts, idx = F.scene_for(seed)
        t0, t1 = 1e-6 * scale, 1e9 * scale
        prod = rayrs_amd.Scene(objs, t0, t1, heur, hdri, device=-1)
        osc = _oracle.OracleScene(objs, t0, t1, heur, hdri).use_walk_tree(prod)
        osg = _oracle.OracleScene(objs, t0, t1, heur, hdri).use_walk_tree(prod, gate=True)
        ref = osc.intersect_batch(o, d, t0, t1, traversal=0)
        default = osc.intersect_batch(o, d, t0, t1, traversal=2)
        probe = osc.cull_margin_probe(o, d, t0, t1)
        try:
            _oracle.set_cull_margin(float("inf"))
            leaves = osc.intersect_batch(o, d, t0, t1, traversal=2)   # the default tree, nothing culled
            exact = osg.intersect_batch(o, d, t0, t1, traversal=2)    # the gate tree, nothing culled
        finally:
            _oracle.set_cull_margin(2.0 ** -10)
        return ref, default, leaves, exact, probe

    # culling
    (rt, robj), (wt, wobj), (lt, lobj), (xt, xobj), (w, in_front, beyond) = walks(
        79, np.array([[0.8461539702186601, -0.3178647511202013, 1.6666324107517303]]),
        np.array([[-4.878144810174007e-10, -0.00017608737629874798, -0.004121392011531156]]))
    assert robj[0] >= 0 and wobj[0] >= 0 and wt[0] > rt[0] * 1.01     # the walk's hit lies 2 % behind the reference's
    assert w > 2.0 ** -10 and beyond >= 1
    assert lobj[0] == robj[0] and lt[0] == rt[0]                      # the leaf boxes are not what loses it
    assert xobj[0] == robj[0] and xt[0] == rt[0]
    # leaf boxes
    (rt, robj), (wt, wobj), (lt, lobj), (xt, xobj), _ = walks(
        2, np.array([[-2.8981278659447747, -633.8240400572021, 259665.5017321564]]),
        np.array([[0.0002220828721502402, 0.05051414085108857, -20.691401286965466]]))
    assert robj[0] >= 0 and wobj[0] >= 0 and wobj[0] != robj[0] and 0 < wt[0] / rt[0] - 1 < 1e-7
    assert lobj[0] == wobj[0] and lt[0] == wt[0]                      # with or without culling
    assert xobj[0] == robj[0] and xt[0] == rt[0]


def test_the_hot_group_is_the_floors_group_on_the_mesh_scenes_and_absent_on_the_sphere_scenes():
    """scene_host.cpp pick_hot_group: the group with the largest gating box, if that covers a quarter of the root Node's
    box and the scene has at least eight groups.  On the obj scenes that is the group the 50 x 50 floor is in; the
    reference's sphere scenes (a handful of groups: the local-pool route) have none.  The oracle's walk of the tree
    without the group, with the group tested beside it, returns the recursion's hits and tests exactly as many
    primitives as its walk of the whole gate tree (the visit set is the same set)."""
    for fn in (scenes.diffuse_single_sphere, scenes.cook_torrance_spheres_metallic, scenes.material_test):
        cam, objs, heur = fn()
        assert rayrs_amd.Scene(objs, 1e-6, 1e6, heur, HDRI, device=-1).info()["hot_count"] == 0
    for level, light in ((2, False), (3, True)):
        cam, objs, heur = scenes.mesh_scene(level, area_light=light)
        prod = rayrs_amd.Scene(objs, 1e-6, 1e6, heur, HDRI, device=-1)
        info = prod.info()
        assert 1 <= info["hot_count"] <= 4 and info["hot_n_wide"] >= 2
        pp = prod.export_bvh()[2]
        assert objs[0].kind == "plane" and objs[0].umax - objs[0].umin == 50.0  # the floor is the first object
        assert 0 in [int(pp[info["hot_first"] + k]) for k in range(info["hot_count"])]
        o, d = random_rays(20000, 40 + level)
        both_walks(objs, heur, o, d)
        ocam = _oracle.OracleCamera(*scenes.camera_for_resolution(cam, 48, 32))
        og = _oracle.OracleScene(objs, 1e-6, 1e6, heur, HDRI).use_product_walk(prod, hot=False)
        oh = _oracle.OracleScene(objs, 1e-6, 1e6, heur, HDRI).use_product_walk(prod, hot=True)
        fg, sg = og.render(ocam, 4, traversal=2)
        fh, sh = oh.render(ocam, 4, traversal=2)
        fr, sr = og.render(ocam, 4, traversal=0)
        assert np.array_equal(fg.view(np.uint64), fr.view(np.uint64)) and np.array_equal(fh.view(np.uint64), fr.view(np.uint64))
        for k in ("rays", "tri_tests", "sphere_tests", "plane_tests"):
            assert sg[k] == sh[k], k
        assert sh["interior_visits"] <= sg["interior_visits"]
