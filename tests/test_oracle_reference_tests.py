"""The reference's own unit tests and doc-test known answers for the hot path,
re-expressed against the CPU oracle (the reference is Rust and cannot be built
here).  Each test names the reference test it restates; paths are relative to
/root/reference/rayrs-lib/src.  These pin the oracle's intersection layer; the
reference has no numeric test for materials or radiance (parity unpinned there).
"""
import ctypes as C
import math

import numpy as np
import pytest

import _oracle
from rayrs_amd import procedural
from rayrs_amd.api import Axis, BvhHeuristic, Emission, Material, Object

L = _oracle.lib()
HDRI = procedural.make_hdri(32, 16)
NOREFLECT = Material.NoReflect()
DARK = Emission.Dark()


def sphere_t(radius, c, o, d):
    t = C.c_double()
    ok = L.orc_sphere_intersect(radius, _oracle.d3(c), _oracle.d3(o), _oracle.d3(d), C.byref(t))
    return t.value if ok else None


def plane_t(axis, o, d):
    t = C.c_double()
    ok = L.orc_plane_intersect(axis, -1., 1., -1., 1., 0., _oracle.d3(o), _oracle.d3(d), C.byref(t))
    return t.value if ok else None


def aabb(o, d, tmin, tmax):
    box = (C.c_double * 6)(-1., 1., -1., 1., -1., 1.)
    return bool(L.orc_aabb_intersect(box, _oracle.d3(o), _oracle.d3(d), tmin, tmax))


# ---- geometry.rs:739-774 spheres

def test_sphere_new_negative_radius():  # :739-743 should_panic
    with pytest.raises(ValueError):
        _oracle.OracleScene([Object.sphere(-1., (0, 0, 0), NOREFLECT, DARK)], 1e-6, 1e6, BvhHeuristic.Midpoint, HDRI)


def test_intersect_sphere_outside():  # :745-750
    assert sphere_t(1., (0, 0, 0), (0, 0, 5), (0, 0, -1)) > 0.


def test_intersect_sphere_inside():  # :753-758
    assert sphere_t(1., (0, 0, 0), (0, 0, 0), (0, 1, 0)) > 0.


def test_intersect_sphere_miss():  # :761-766
    assert sphere_t(1., (0, 0, 0), (0, 5, 0), (0, 1, 0)) is None


def test_intersect_sphere_glancing():  # :769-774
    assert sphere_t(1., (0, 0, 0), (0.99999, -5., 0.), (0, 1, 0)) > 0.


# ---- geometry.rs:776-845 planes

def test_plane_new_min_max_order():  # :776-780 should_panic
    with pytest.raises(ValueError):
        _oracle.OracleScene([Object.plane(Axis.Z, 1., -1., 1., -1., 0., NOREFLECT, DARK)], 1e-6, 1e6,
                            BvhHeuristic.Midpoint, HDRI)


@pytest.mark.parametrize("axis,o,d", [
    (Axis.X, (5, 0, 0), (-1, 0, 0)),   # test_intesect_x_plane_front :783
    (Axis.X, (-5, 0, 0), (1, 0, 0)),   # test_intesect_x_plane_back  :791
    (Axis.Y, (0, 5, 0), (0, -1, 0)),   # :799
    (Axis.Y, (0, -5, 0), (0, 1, 0)),   # :807
    (Axis.Z, (0, 0, 5), (0, 0, -1)),   # :815
    (Axis.Z, (0, 0, -5), (0, 0, 1)),   # :823
])
def test_intersect_plane_front_and_back(axis, o, d):
    assert plane_t(axis, o, d) > 0.


# ---- geometry.rs:847-887 AABB

def test_aabb_intersection_outside_x():  # :848
    assert aabb((-5, 0, 0), (1, 0, 0), 0.001, 1000.)


def test_aabb_intersection_outside_y():  # :855
    assert aabb((0, -5, 0), (0, 1, 0), 0.0001, 1000.)


def test_aabb_intersection_outside_z():  # :862
    assert aabb((0, 0, -5), (0, 0, 1), 0.001, 1000.)


def test_aabb_intersection_inside():  # :869
    assert aabb((0, 0, 0), (0, 0, 1), 0.001, 1000.)


def test_aabb_intersection_miss():  # :876
    assert not aabb((1.1, 0, 0), (0, 1, 1), 0.001, 1000.)


def test_aabb_intersection_miss2():  # :883
    assert not aabb((2, 0, 0), (-1, -2, 0), 0.001, 1000.)


# ---- bvh.rs:543-559

def test_bvh_intersect_node_leafnode():
    """Node(bbox, [Leaf(unit sphere)]), ray (-5,0,0)->(1,0,0), range (0.001, 1000): t == 4.0 exactly."""
    sc = _oracle.OracleScene([Object.sphere(1., (0, 0, 0), NOREFLECT, DARK)], 1e-6, 1e6, BvhHeuristic.Midpoint, HDRI)
    for trav in (0, 1):
        obj, t = sc.intersect((-5., 0., 0.), (1., 0., 0.), 0.001, 1000., traversal=trav)
        assert obj == 0 and t == 4.0


# ---- doc-test scalars

def test_camera_pixels_doc_test():  # lib.rs:141-151, :163-173
    cam = _oracle.OracleCamera((1, 1, 1), (0, 1, 0), (0, 0, 0), 90., 20., 10., 90)
    assert cam.x_pixels() == 4580 and cam.y_pixels() == 2290


def test_camera_asserts():  # lib.rs:108-111
    for args in [((1, 1, 1), (0, 1, 0), (0, 0, 0), 0., 20., 10., 90), ((1, 1, 1), (0, 1, 0), (0, 0, 0), 180., 20., 10., 90),
                 ((1, 1, 1), (0, 1, 0), (0, 0, 0), 90., 0., 10., 90), ((1, 1, 1), (0, 1, 0), (0, 0, 0), 90., 20., -1., 90),
                 ((1, 1, 1), (0, 1, 0), (1, 1, 1), 90., 20., 10., 90)]:
        with pytest.raises(ValueError):
            _oracle.OracleCamera(*args)


def test_bbox_doc_tests():
    """Two unit spheres at x = -1 and x = +1: xmin -2, xmax 2 (geometry.rs:541-542), centre 0 (:575),
    volume 16 (:607), surface area 40 (:638)."""
    sc = _oracle.OracleScene([Object.sphere(1., (-1, 0, 0), NOREFLECT, DARK), Object.sphere(1., (1, 0, 0), NOREFLECT, DARK)],
                             1e-6, 1e6, BvhHeuristic.Midpoint, HDRI)
    box, cen, vol, sa = sc.bbox()
    assert box[0] == -2. and box[1] == 2.
    assert cen == [0., 0., 0.]
    assert vol == 16. and sa == 40.


def test_scene_new_asserts():  # lib.rs:234-235, bvh.rs:229
    s = Object.sphere(1., (0, 0, 0), NOREFLECT, DARK)
    with pytest.raises(ValueError):
        _oracle.OracleScene([s], -1., 1e6, BvhHeuristic.Midpoint, HDRI)
    with pytest.raises(ValueError):
        _oracle.OracleScene([s], 1., 1., BvhHeuristic.Midpoint, HDRI)
    with pytest.raises(ValueError):
        _oracle.OracleScene([], 1e-6, 1e6, BvhHeuristic.Midpoint, HDRI)


def test_material_ctor_asserts():  # material.rs:609, :706-708, :833-837, :1068-1072
    bad = [Material.LambertianDiffuse((1.5, 0, 0)), Material.CookTorranceGlass((1, 1, 1), 0., 1.45),
           Material.CookTorranceGlass((1, 1, 1), 0.1, -1.), Material.Glass((1, 1, 1), float("inf")),
           Material.Plastic((0.5, 0.5, 0.5), (2, 1, 1), 0.1, 1.45)]
    for m in bad:
        with pytest.raises(ValueError):
            _oracle.OracleScene([Object.sphere(1., (0, 0, 0), m, DARK)], 1e-6, 1e6, BvhHeuristic.Midpoint, HDRI)
    with pytest.raises(ValueError):
        _oracle.OracleScene([Object.sphere(1., (0, 0, 0), NOREFLECT, Emission.new(-1., (1, 1, 1)))], 1e-6, 1e6,
                            BvhHeuristic.Midpoint, HDRI)


# ---- vecmath.rs doc tests that the hot path relies on

def test_powf_doc_test_value():  # vecmath.rs:363-366 -- f64::powf is the platform pow
    assert 0.5 ** (1 / 2.2) == 0.7297400528407231


def test_triangle_normal_matches_disabled_bvh_test_debug_string():
    """bvh.rs:540 (disabled test) prints Triangle::new((-1,-1,-.5), (-.5,-1,-1), (1,1,-1.5)): e1 (0.5,0,-0.5),
    e2 (2,2,-1), normal (2/3,-1/3,2/3), area 0.75."""
    n = (C.c_double * 3)()
    L.orc_triangle_normal(_oracle.d3((-1., -1., -0.5)), _oracle.d3((-0.5, -1., -1.)), _oracle.d3((1., 1., -1.5)), n)
    assert list(n) == [0.6666666666666666, -0.3333333333333333, 0.6666666666666666]


# ---- vecmath.rs:812-893: the reference's eleven vector tests, same operands, same `==`

def test_vec_add():  # :817-822
    assert _oracle.vec_op("add", (1., 2., 3.), (2., 4., 6.)) == (3., 6., 9.)


def test_vec_sub():  # :825-830
    assert _oracle.vec_op("sub", (4., 3., 2.), (1., 1., 1.)) == (3., 2., 1.)


def test_vec_mul():  # :833-838
    assert _oracle.vec_op("mul", (1., 4., 8.), (2., 2., 2.)) == (2., 8., 16.)


def test_vec_scalar_mul_both_orders():  # :841-853 (3. * v and v * 3.: one implementation, vecmath.rs:620-650)
    assert _oracle.vec_op("scale", (1., 2., 3.), s=3.) == (3., 6., 9.)


def test_vec_dot():  # :856-859
    assert _oracle.vec_dot((1., 2., 3.), (1., 2., 3.)) == 14.


def test_vec_cross1():  # :862-867
    assert _oracle.vec_op("cross", (1., 0., 0.), (0., 1., 0.)) == (0., 0., 1.)


def test_vec_cross2():  # :870-875
    assert _oracle.vec_op("cross", (1., 0., 0.), (0., 0., 1.)) == (0., -1., 0.)


def test_vec_mag2():  # :878-881
    assert _oracle.vec_mag2((1., 2., 3.)) == 14.


def test_vec_pow():  # :884-887
    assert _oracle.vec_op("powf", (1., 2., 3.), s=2.) == (1., 4., 9.)


def test_vec_clip():  # :890-893
    assert _oracle.vec_op("clip", (-1., 2., 0.5), s=0., t=1.) == (0., 1., 0.5)


def test_div_multiplies_by_the_reciprocal_but_div_assign_divides():  # vecmath.rs:690-714
    a = (1., 7., 49.)
    assert _oracle.vec_op("div", a, s=49.) == tuple(x * (1. / 49.) for x in a)
    assert _oracle.vec_op("div_assign", a, s=49.) == tuple(x / 49. for x in a)
    assert _oracle.vec_op("div", a, s=49.) != _oracle.vec_op("div_assign", a, s=49.)  # 49 * (1/49) != 1


def test_orthonormal_basis_doc_test():  # vecmath.rs:333-339: x.cross(y) == z for z = unit_z
    z = (0., 0., 1.)
    x, y = _oracle.orthonormal_basis(z)
    assert _oracle.vec_op("cross", x, y) == z
    for n in ((1., 0., 0.), (0., 1., 0.), (0., -1., 0.)):   # both branches of :343-347
        e1, e2 = _oracle.orthonormal_basis(n)
        assert abs(_oracle.vec_dot(e1, n)) < 1e-15 and abs(_oracle.vec_dot(e2, n)) < 1e-15
        assert max(abs(a - b) for a, b in zip(_oracle.vec_op("cross", e1, e2), n)) < 1e-15


def test_aabb_expand_equals_from_object_list_doc_test():  # geometry.rs:655-673
    objs = [Object.sphere(1., (-1., 0., 0.), NOREFLECT, DARK), Object.sphere(1., (1., 0., 0.), NOREFLECT, DARK)]
    boxes = [_oracle.OracleScene([o], 1e-6, 1e6, BvhHeuristic.Midpoint, HDRI).bbox()[0] for o in objs]
    both = _oracle.OracleScene(objs, 1e-6, 1e6, BvhHeuristic.Midpoint, HDRI).bbox()[0]
    assert _oracle.aabb_expand(boxes[0], boxes[1]) == both == [-2., 2., -1., 1., -1., 1.]
