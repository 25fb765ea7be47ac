"""The reference's own unit vectors run straight through the HIP path (missing 5): every case of
rayrs-lib/src/geometry.rs:735-888 and bvh.rs:543-559 as a BVH query on the GPU (rayrs_test_intersect: the traversal
kernel's own walk and primitive tests), on the smallest scene that holds the case's primitive.  The same vectors pin the
CPU oracle in tests/test_oracle_reference_tests.py; here the kernels answer them themselves instead of inheriting the
answer through the oracle.  Camera::new's doc-test scalars (lib.rs:141-177: 4580 x 2290) go through rayrs_camera_new in
tests/test_abi.py::test_camera_matches_reference_doc_test (host code, no GPU)."""
import numpy as np
import pytest

import rayrs_amd
from rayrs_amd import _ffi, procedural
from rayrs_amd.api import Axis, BvhHeuristic, Emission, Material, Object

pytestmark = pytest.mark.gpu

HDRI = procedural.make_hdri(32, 16)
NR, DARK = Material.NoReflect(), Emission.Dark()


def query(objs, o, d, t0, t1, exact):
    """(t, object index or -1) of one BVH query on the device."""
    scene = rayrs_amd.Scene(objs, t0, t1, BvhHeuristic.Midpoint, HDRI, device=0)
    oo = np.array([o], dtype=np.float64)
    dd = np.array([d], dtype=np.float64)
    t = np.zeros(1)
    obj = np.zeros(1, dtype=np.int64)
    _ffi.check(scene._L.rayrs_test_intersect(scene._h, oo.ctypes.data, dd.ctypes.data, 1, int(exact), t.ctypes.data,
                                             obj.ctypes.data), "rayrs_test_intersect")
    return float(t[0]), int(obj[0])


UNIT_SPHERE = [Object.sphere(1., (0., 0., 0.), NR, DARK)]


@pytest.mark.parametrize("exact", [1, 0])
@pytest.mark.parametrize("o,d,hit", [
    ((0., 0., 5.), (0., 0., -1.), True),          # test_intersect_sphere_outside  geometry.rs:745-750
    ((0., 0., 0.), (0., 1., 0.), True),           # test_intersect_sphere_inside   :753-758
    ((0., 5., 0.), (0., 1., 0.), False),          # test_intersect_sphere_miss     :761-766
    ((0.99999, -5., 0.), (0., 1., 0.), True),     # test_intersect_sphere_glancing :769-774
])
def test_sphere_cases(o, d, hit, exact):
    t, obj = query(UNIT_SPHERE, o, d, 1e-6, 1e6, exact)
    assert (obj == 0 and t > 0.) if hit else obj == -1


@pytest.mark.parametrize("exact", [1, 0])
@pytest.mark.parametrize("axis,o,d", [
    (Axis.X, (5., 0., 0.), (-1., 0., 0.)),   # test_intesect_x_plane_front geometry.rs:783
    (Axis.X, (-5., 0., 0.), (1., 0., 0.)),   # test_intesect_x_plane_back  :791
    (Axis.Y, (0., 5., 0.), (0., -1., 0.)),   # :799
    (Axis.Y, (0., -5., 0.), (0., 1., 0.)),   # :807
    (Axis.Z, (0., 0., 5.), (0., 0., -1.)),   # :815
    (Axis.Z, (0., 0., -5.), (0., 0., 1.)),   # :823
])
def test_plane_front_and_back(axis, o, d, exact):
    """Plane::new(axis, -1, 1, -1, 1, 0).  Alone in a scene the rectangle sits behind its own zero-thickness box, which
    the slab test never passes (the reference cannot hit it through its BVH either); a small sphere
    well off the ray shares its bottom Node and gives that Node's box a thickness, as any real scene does."""
    objs = [Object.plane(axis, -1., 1., -1., 1., 0., NR, DARK), Object.sphere(0.1, (0.7, 0.7, 0.7), NR, DARK)]
    t, obj = query(objs, o, d, 1e-6, 1e6, exact)
    assert obj == 0 and t == 5.0
    t, obj = query(objs[:1], o, d, 1e-6, 1e6, exact)
    assert obj == -1   # quirk (a), on the device too


@pytest.mark.parametrize("exact", [1, 0])
@pytest.mark.parametrize("o,d,t0,hit", [
    ((-5., 0., 0.), (1., 0., 0.), 0.001, True),     # test_aabb_intersection_outside_x geometry.rs:848
    ((0., -5., 0.), (0., 1., 0.), 0.0001, True),    # outside_y :855
    ((0., 0., -5.), (0., 0., 1.), 0.001, True),     # outside_z :862
    ((0., 0., 0.), (0., 0., 1.), 0.001, True),      # inside    :869
    ((1.1, 0., 0.), (0., 1., 1.), 0.001, False),    # miss      :876
    ((2., 0., 0.), (-1., -2., 0.), 0.001, False),   # miss2     :883
])
def test_aabb_cases_through_the_root_box(o, d, t0, hit, exact):
    """AxisAlignedBoundingBox::new(-1, 1, -1, 1, -1, 1) is the unit sphere's box, the root Node's box of this scene, with
    the test's own t range as the scene's: a query gets past it exactly in the reference's `true` cases (and then finds
    the sphere: each of those rays runs through the centre)."""
    t, obj = query(UNIT_SPHERE, o, d, t0, 1000., exact)
    assert (obj == 0 and t > 0.) if hit else obj == -1


@pytest.mark.parametrize("exact", [1, 0])
def test_bvh_intersect_node_leafnode(exact):
    """bvh.rs:543-559: Node(bbox, [Leaf(unit sphere)]), ray (-5,0,0) -> (1,0,0), range (0.001, 1000): t == 4.0."""
    t, obj = query(UNIT_SPHERE, (-5., 0., 0.), (1., 0., 0.), 0.001, 1000., exact)
    assert obj == 0 and t == 4.0
