"""Reference behaviours that change results and must be kept,
pinned on the oracle.  Each is also covered on the GPU by the bit-exact frame tests."""
import ctypes as C
import math

import numpy as np

import _oracle
import rayrs_amd
from rayrs_amd import procedural
from rayrs_amd.api import Axis, BvhHeuristic, Emission, Fresnel, Material, Object

L = _oracle.lib()
HDRI = procedural.make_hdri(32, 16)
NR, DARK = Material.NoReflect(), Emission.Dark()
MID = BvhHeuristic.Midpoint


def test_a_zero_thickness_node_box_is_unhittable():
    """AABB `tmax <= tmin -> miss` (geometry.rs:474/491/508): a Node whose children are coplanar
    axis-aligned rectangles has a zero-thickness box that no ray can enter."""
    box = (C.c_double * 6)(-1., 1., 0., 0., -1., 1.)
    assert not L.orc_aabb_intersect(box, _oracle.d3((0, 5, 0)), _oracle.d3((0, -1, 0)), 1e-6, 1e6)
    floor = Object.plane(Axis.Y, -1., 1., -1., 1., 0., NR, DARK)
    sc = _oracle.OracleScene([floor], 1e-6, 1e6, MID, HDRI)  # Node(bbox(floor), [Leaf(floor)])
    for trav in (0, 1):
        assert sc.intersect((0, 5, 0), (0, -1, 0), 1e-6, 1e6, trav)[0] == -1
    # as soon as the node box has thickness the floor is hit again
    sc = _oracle.OracleScene([floor, Object.sphere(0.5, (5, 3, 0), NR, DARK)], 1e-6, 1e6, MID, HDRI)
    for trav in (0, 1):
        assert sc.intersect((0, 5, 0), (0, -1, 0), 1e-6, 1e6, trav) == (0, 5.0)


def test_b_sphere_picks_t1_below_tmin_and_loses_the_far_side():
    """geometry.rs:116-128 + bvh.rs:406: a ray leaving the surface with 0 <= t1 < tmin gets Some(t1),
    which the leaf range test then rejects -- the far side is not reported."""
    o = (0.0, 0.0, 1.0 - 1e-9)  # just inside the surface, heading inwards through the sphere
    t = C.c_double()
    assert L.orc_sphere_intersect(1.0, _oracle.d3((0, 0, 0)), _oracle.d3(o), _oracle.d3((0, 0, -1)), C.byref(t))
    sc = _oracle.OracleScene([Object.sphere(1.0, (0, 0, 0), NR, DARK)], 1e-6, 1e6, MID, HDRI)
    o2 = (0.0, 0.0, 1.0 + 5e-7)  # outside by less than tmin
    assert L.orc_sphere_intersect(1.0, _oracle.d3((0, 0, 0)), _oracle.d3(o2), _oracle.d3((0, 0, -1)), C.byref(t))
    assert 0.0 <= t.value < 1e-6
    assert sc.intersect(o2, (0, 0, -1), 1e-6, 1e6, 0)[0] == -1  # t2 ~ 2.0 is never considered


def test_c_degenerate_triangle_gives_nan_and_is_rejected_by_the_leaf():
    """geometry.rs:364-374: no determinant guard; Some(NaN) / inf fail `t > tmin && t < tmax`."""
    t = C.c_double()
    p = [(0., 0., 0.), (1., 0., 0.), (2., 0., 0.)]  # zero area
    ok = L.orc_triangle_intersect(_oracle.d3(p[0]), _oracle.d3(p[1]), _oracle.d3(p[2]), _oracle.d3((0.5, 1, 0)),
                                  _oracle.d3((0, -1, 0)), C.byref(t))
    assert ok and math.isnan(t.value)
    sc = _oracle.OracleScene([Object.triangle(*p, NR, DARK), Object.sphere(0.25, (0, 5, 0), NR, DARK)], 1e-6, 1e6, MID,
                             HDRI)
    for trav in (0, 1):
        assert sc.intersect((0.5, 1, 0), (0, -1, 0), 1e-6, 1e6, trav)[0] == -1


def test_back_faces_are_hit():
    t = C.c_double()
    p = [(-1., 0., 0.), (1., 0., 0.), (0., 1., 0.)]
    for d in ((0, 0, -1), (0, 0, 1)):
        o = (0.0, 0.3, -5.0 * d[2])
        assert L.orc_triangle_intersect(*[_oracle.d3(x) for x in p], _oracle.d3(o), _oracle.d3(d), C.byref(t))
        assert t.value == 5.0


def test_d_noscatter_at_an_emitter_drops_its_emission():
    """lib.rs:550: NoScatter returns `light` gathered so far; the emitter's own emission is only
    added on Scatter (lib.rs:534)."""
    light_nr = Object.plane(Axis.YRev, -5., 5., -5., 5., 4., NR, Emission.new(5., (1, 1, 1)))
    light_diffuse = Object.plane(Axis.YRev, -5., 5., -5., 5., 4., Material.LambertianDiffuse((0.5, 0.5, 0.5)),
                                 Emission.new(5., (1, 1, 1)))
    pad = Object.sphere(0.1, (9, 9, 9), NR, DARK)  # gives the node box thickness
    for lamp, expect_light in ((light_nr, False), (light_diffuse, True)):
        sc = _oracle.OracleScene([lamp, pad], 1e-6, 1e6, MID, HDRI)
        rgb, rays, _ = sc.radiance((0, 0, 0), (0, 1, 0), 50, key=12345)
        assert (rgb.min() >= 5.0) == expect_light


def test_e_f_throughput_division_and_exhausted_budget():
    """lib.rs:559: when the bounce budget runs out `light` is returned without a background term."""
    mirror = Material.Reflect((1, 1, 1))
    corridor = [Object.plane(Axis.Y, -50., 50., -50., 50., 0., mirror, DARK),
                Object.plane(Axis.YRev, -50., 50., -50., 50., 2., mirror, DARK)]  # two facing mirrors
    sc = _oracle.OracleScene(corridor, 1e-6, 1e6, MID, HDRI)
    rgb, rays, _ = sc.radiance((0, 1, 0), (0.01, 1, 0.02), 7, key=1)
    assert rays == 7 and np.all(rgb == 0.0)   # throughput stays 1: roulette never fires, budget runs out
    rgb, rays, _ = sc.radiance((0, 1, 0), (0.01, 1, 0.02), 5000, key=1)
    assert rays < 5000 and np.all(rgb > 0.0)  # with budget left the path walks out and sees the sky
    # (j) seen from behind, a Reflect surface gives a negative colour (signed n.l) and the path dies
    ball = _oracle.OracleScene([Object.sphere(100.0, (0, 0, 0), mirror, DARK)], 1e-6, 1e6, MID, HDRI)
    rgb, rays, _ = ball.radiance((0, 0, 0), (0.3, 1, 0.2), 7, key=1)
    assert rays == 1 and np.all(rgb == 0.0)


def test_g_camera_focal_length_and_pixel_shift():
    """lib.rs:131: z = width / tan(fov/2) * z_hat (not width/2/tan); main.rs:74-75: camera rows run
    1..=height, not 0..height."""
    cam = _oracle.OracleCamera((0, 0, 0), (0, 1, 0), (0, 0, 1), 90., 2., 1., 100)
    assert abs(cam.desc.z[2] - 2.0) < 1e-12  # width / tan(45 deg)
    o, d, draws = cam.primary_ray(cam.y_pixels(), cam.x_pixels(), key=99)
    assert draws == 2
    # pixel index == pixel count: the jittered sample lies beyond the far edge of the sensor
    assert d[0] >= 1.0 - 1e-12 or d[0] <= -1.0 + 1e-12 or abs(d[1]) >= 0.5 - 1e-12


def test_h_background_is_black_at_integral_texel_coordinates():
    cam_args = None
    sc = _oracle.OracleScene([Object.sphere(1.0, (0, 0, 0), NR, DARK)], 1e-6, 1e6, MID, HDRI)
    rgb = sc.background([[0.0, 1.0, 0.0], [0.3, 0.4, 0.5]])
    assert np.all(rgb[0] == 0.0)      # theta == 0 -> y integral -> all four weights zero (lib.rs:265-284)
    assert np.all(rgb[1] > 0.0)


def test_i_beckmann_pdf_is_d_of_h_and_tan_guard_never_fires():
    m = Material.CookTorrance((1, 1, 1), 0.3, Fresnel.SchlickMetallic((0.8, 0.8, 0.8)))
    n = np.array([[0.0, 1.0, 0.0]] * 64)
    v = np.array([[0.0, 1.0, 0.0]] * 64)
    key = np.arange(64, dtype=np.uint64) * 7919 + 1
    sc, col, dr, nd = _oracle.material_evaluate(m, n, v, key)
    assert np.all(nd == 2)            # phi then u (material.rs:1009-1011)
    assert sc.sum() > 32 and np.all(np.isfinite(col[sc == 1]))


def test_j_dielectrics_flip_the_normal_others_do_not():
    n = np.array([[0.0, 1.0, 0.0]])
    v_below = np.array([[0.0, -1.0, 0.0]])  # viewer behind the surface
    key = np.array([5], dtype=np.uint64)
    sc, col, dr, nd = _oracle.material_evaluate(Material.Reflect((1, 1, 1)), n, v_below, key)
    assert sc[0] == 1 and np.allclose(dr[0], [0, -1, 0]) and np.allclose(col[0], [-1, -1, -1])  # signed n.l
    sc, col, dr, nd = _oracle.material_evaluate(Material.Refract((1, 1, 1), 1.45), n, v_below, key)
    assert sc[0] == 1 and np.allclose(dr[0], [0, 1, 0]) and np.all(col[0] > 0)


def test_draw_counts_per_material():
    """RNG call sites in program order: Cosine 2, Beckmann 2, Glass 0/1, CTGlass 2/3, Plastic 3."""
    r = np.random.default_rng(2)
    n = r.normal(size=(400, 3))
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    v = n + 0.3 * r.normal(size=(400, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    key = r.integers(0, 2 ** 63, 400, dtype=np.uint64)
    draws = lambda m: set(_oracle.material_evaluate(m, n, v, key)[3].tolist())
    assert draws(Material.LambertianDiffuse((0.5, 0.5, 0.5))) == {2}
    assert draws(Material.Reflect((0.5, 0.5, 0.5))) == {0}
    assert draws(Material.CookTorrance((1, 1, 1), 0.2, Fresnel.SchlickMetallic((0.8, 0.8, 0.8)))) == {2}
    assert draws(Material.Glass((1, 1, 1), 1.45)) <= {0, 1}
    assert draws(Material.CookTorranceGlass((1, 1, 1), 0.2, 1.45)) <= {2, 3}
    assert draws(Material.Plastic((0.5, 0.5, 0.5), (1, 1, 1), 0.2, 1.45)) == {3}
    assert draws(Material.NoReflect()) == {0}


def test_sentinel_boxes_of_the_wide_records():
    """The kernel's wide records give the two kinds of slot the reference never box-tests a
    box instead of a special case (scene_host.cpp): all of space for a direct leaf, the inverted
    box for an unused slot.  Under AxisAlignedBoundingBox::intersect (geometry.rs:458-513, as
    restated by the oracle) the first is entered by every ray and the second by none -- for
    axis-parallel directions (1/0 = inf) and negative ones too."""
    import ctypes as C
    import numpy as np
    L = _oracle.lib()
    inf = float("inf")
    everything = (C.c_double * 6)(-inf, inf, -inf, inf, -inf, inf)
    nothing = (C.c_double * 6)(inf, -inf, inf, -inf, inf, -inf)
    r = np.random.default_rng(2)
    dirs = [(1, 0, 0), (0, -1, 0), (0, 0, 1), (-1, -1, -1), (1e-300, 1, 0), (0.0, -0.0, 1.0)]
    dirs += [tuple(v) for v in r.normal(size=(50, 3))]
    for d in dirs:
        for o in [(0, 0, 0), (3.5, -2e6, 7e-9), (-1e300, 1e300, 0)]:
            assert L.orc_aabb_intersect(everything, _oracle.d3(o), _oracle.d3(d), 1e-6, 1e6) == 1
            assert L.orc_aabb_intersect(nothing, _oracle.d3(o), _oracle.d3(d), 1e-6, 1e6) == 0


def test_kernel_walk_equals_reference_on_degenerate_rays():
    """Rays built to hit the slab test's special cases -- origins exactly on box planes, direction
    components exactly zero (1/0 = inf, 0 * inf = NaN, which f64::max/min ignore, geometry.rs:458-513)
    -- through a scene on an integer grid, so that such coincidences are the rule.  The walk over
    the four-slot records of the product's walk tree (what the kernel does) must find what the reference's recursion
    finds: skipping a parent's box in favour of its child's is only sound if the slab test is
    monotone in these cases too."""
    r = np.random.default_rng(11)
    nr, dark = Material.NoReflect(), Emission.Dark()
    objs = []
    for i in range(400):
        c = r.integers(-6, 7, 3).astype(float)
        if i % 3 == 0:
            objs.append(Object.sphere(float(r.integers(1, 3)) * 0.5, c, nr, dark))
        elif i % 3 == 1:
            objs.append(Object.plane(int(r.integers(0, 6)), c[0], c[0] + 2.0, c[1], c[1] + 1.0, c[2], nr, dark))
        else:
            objs.append(Object.triangle(c, c + r.integers(-2, 3, 3), c + r.integers(-2, 3, 3), nr, dark))
    n = 4000
    o = r.integers(-8, 9, (n, 3)).astype(float)
    o[n // 2:] += r.integers(0, 2, (n - n // 2, 3)) * 0.5
    d = r.integers(-2, 3, (n, 3)).astype(float)
    d[(d == 0).all(axis=1)] = (1.0, 0.0, 0.0)
    d[::7, 1] = -0.0  # negative zero: 1/-0 = -inf
    for heur in (BvhHeuristic.Sah(1000), BvhHeuristic.Midpoint):
        hdri = procedural.make_hdri(32, 16)
        osc = _oracle.OracleScene(objs, 1e-6, 1e6, heur, hdri)
        osc.use_walk_tree(rayrs_amd.Scene(objs, 1e-6, 1e6, heur, hdri, device=-1))  # the product's tree, host only
        t0, obj0 = osc.intersect_many(o, d, 1e-6, 1e6, traversal=0)
        assert (obj0 >= 0).sum() > 500
        for trav in (1, 2):
            t, obj = osc.intersect_many(o, d, 1e-6, 1e6, traversal=trav)
            assert np.array_equal(obj, obj0), trav
            assert np.array_equal(t.view(np.uint64), t0.view(np.uint64)), trav
