"""GPU parity, function by function: each piece of the hot path is run on the
MI355X through the C ABI's self-test hooks and compared with the CPU oracle on
the same seeded inputs.  Bar: bit-exact (the kernel and the oracle share the
numeric contract of include/rayrs_numeric.h and evaluate the reference's
expressions in the same order)."""
import math

import numpy as np
import pytest

import _oracle
import ctypes as C
from rayrs_amd import _ffi, procedural, scenes
from rayrs_amd.api import Fresnel, Material, Scene

pytestmark = pytest.mark.gpu

HDRI = procedural.make_hdri(256, 128)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def gpu_math(fn, x, y=None):
    L = _ffi.lib()
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.zeros_like(x)
    yp = None
    if y is not None:
        y = np.ascontiguousarray(y, dtype=np.float64)
        yp = y.ctypes.data
    _ffi.check(L.rayrs_test_math(0, fn, x.ctypes.data, yp, len(x), out.ctypes.data), "rayrs_test_math")
    return out


MATH_CASES = [
    ("sin", 0, lambda r, n: r.uniform(0, 2 * math.pi, n)),
    ("cos", 1, lambda r, n: r.uniform(0, 2 * math.pi, n)),
    ("tan", 2, lambda r, n: np.concatenate([r.uniform(0, math.pi, n // 2),
                                            math.pi / 2 + r.uniform(-1e-7, 1e-7, n - n // 2)])),
    ("log", 3, lambda r, n: np.concatenate([1 - r.uniform(0, 1, n // 2), 2.0 ** -r.uniform(0, 60, n - n // 2)])),
    ("exp", 4, lambda r, n: np.concatenate([-r.uniform(0, 60, n // 2), -10 ** r.uniform(-3, 3.2, n - n // 2)])),
    ("acos", 5, lambda r, n: np.concatenate([r.uniform(-1, 1, n // 2), 1 - 10 ** -r.uniform(0, 17, n - n // 2)])),
    ("sqrt", 7, lambda r, n: 10 ** r.uniform(-30, 30, n)),
]


@pytest.mark.parametrize("name,fn,gen", MATH_CASES, ids=[c[0] for c in MATH_CASES])
def test_elementary_functions_bit_exact(name, fn, gen):
    x = gen(np.random.default_rng(fn + 1), 20000)
    special = {5: [1.0, -1.0, 0.0, 1.0000000000000002, float("nan")], 3: [1.0, 2.0 ** -53, 0.0],
               4: [0.0, -745.2, -800.0, -708.5, -1e300], 2: [math.acos(0.0), 0.0, math.pi]}
    x = np.concatenate([x, np.array(special.get(fn, []), dtype=np.float64)])
    assert np.array_equal(bits(gpu_math(fn, x)), bits(_oracle.math_fn(fn, x)))


def test_atan2_and_division_bit_exact():
    r = np.random.default_rng(9)
    y, x = r.normal(size=20000), r.normal(size=20000)
    y[:8] = [0.0, 0.0, 1.0, -1.0, 0.0, -0.0, 1e-300, 1e300]
    x[:8] = [1.0, -1.0, 0.0, 0.0, 0.0, -1.0, 1e300, 1e-300]
    assert np.array_equal(bits(gpu_math(6, y, x)), bits(_oracle.math_fn(6, y, x)))
    a, b = r.normal(size=20000) * 10 ** r.uniform(-100, 100, 20000), r.normal(size=20000)
    assert np.array_equal(bits(gpu_math(8, a, b)), bits(a / b))


def test_three_quotients_by_one_denominator_are_the_three_divisions():
    """triangle_intersect's three true divisions (geometry.rs:364-374) share the refined reciprocal of their
    denominator when the hardware scales it alike for all three numerators (device_path.h div3_by).  Whatever
    the operands -- ordinary, subnormal, huge, zero, infinite, NaN, exponents 600 binades apart so that the
    numerators scale the denominator differently -- each quotient must be the IEEE quotient, bit for bit."""
    r = np.random.default_rng(21)
    n = 60000
    a = r.normal(size=n) * 2.0 ** r.integers(-1060, 1020, n)
    b = r.normal(size=n) * 2.0 ** r.integers(-1060, 1020, n)
    special = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 5e-324, -5e-324, 2.2250738585072014e-308,
                        1.7976931348623157e308, 1.0, -1.0, 3.0, 1e-300, 1e300])
    k = len(special)
    a[:k * k] = np.repeat(special, k)
    b[:k * k] = np.tile(special, k)
    a[k * k:k * k + 2000] = r.normal(size=2000)            # the ordinary case: nothing is scaled
    b[k * k:k * k + 2000] = r.normal(size=2000)
    with np.errstate(all="ignore"):
        want = (a / b, (a * 2.0 ** -600) / b, (-3.0 * a) / b)
    for fn, w in zip((9, 10, 11), want):
        got = gpu_math(fn, a, b)
        nan = np.isnan(w)
        assert np.array_equal(np.isnan(got), nan)
        assert np.array_equal(bits(got)[~nan], bits(w)[~nan]), fn


def test_rng_bit_exact():
    L = _ffi.lib()
    r = np.random.default_rng(3)
    n = 5000
    pixel = r.integers(0, 2 ** 22, n, dtype=np.uint64)
    sample = r.integers(0, 4096, n, dtype=np.uint64)
    draw = r.integers(0, 200, n, dtype=np.uint32)
    out = np.zeros(n, dtype=np.uint64)
    seed = 0x5EED
    _ffi.check(L.rayrs_test_rng(0, seed, pixel.ctypes.data, sample.ctypes.data, draw.ctypes.data, n,
                                out.ctypes.data), "rayrs_test_rng")
    ref = np.array([_oracle.rng_bits(seed, int(p), int(s), int(d)) for p, s, d in zip(pixel, sample, draw)],
                   dtype=np.uint64)
    assert np.array_equal(out, ref)


def _rays(n, seed, spread=6.0):
    r = np.random.default_rng(seed)
    o = r.uniform(-spread, spread, (n, 3))
    o[:, 1] = np.abs(o[:, 1]) + 0.05
    target = r.uniform(-2.5, 2.5, (n, 3))
    target[:, 1] = np.abs(target[:, 1])
    d = target - o
    d[: n // 4] = r.normal(size=(n // 4, 3))  # some unnormalised random directions
    d[n // 4: n // 4 + 8, 0] = 0.0            # axis-parallel components: 1/0 = inf slabs
    d[n // 4 + 8: n // 4 + 16, 1] = 0.0
    return np.ascontiguousarray(o), np.ascontiguousarray(d)


def _object_soup():
    """Mixed primitives, coincident centres (median fallback), and five-object splits whose
    left side is a direct leaf: the wide records then hold all-of-space slots and unused ones."""
    from rayrs_amd.api import Emission, Object
    r = np.random.default_rng(4)
    nr, dark = Material.NoReflect(), Emission.Dark()
    objs = []
    for i in range(300):
        c = r.uniform(-2.5, 2.5, 3)
        c[1] = abs(c[1])
        if i % 3 == 0:
            objs.append(Object.sphere(float(r.uniform(0.05, 0.4)), c, nr, dark))
        elif i % 3 == 1:
            objs.append(Object.plane(int(r.integers(0, 6)), c[0], c[0] + 0.5, c[1], c[1] + 0.7, c[2], nr, dark))
        else:
            objs.append(Object.triangle(c, c + r.uniform(-1, 1, 3), c + r.uniform(-1, 1, 3), nr, dark))
    objs += [Object.sphere(0.3, (1.0, 1.0, 1.0), nr, dark) for _ in range(9)]
    return None, objs, scenes.SAH_1000


def _five_with_single_leaf():
    from rayrs_amd.api import Emission, Object
    objs = [Object.sphere(0.5, (float(x), 1.0, 0.0), Material.NoReflect(), Emission.Dark()) for x in (-2, 0.5, 1, 1.5, 2)]
    return None, objs, scenes.SAH_1000


SCENES = {
    "object_soup": _object_soup,
    "five_with_single_leaf": _five_with_single_leaf,
    "single_sphere": lambda: scenes.diffuse_single_sphere(),
    "sphere_row": lambda: scenes.cook_torrance_spheres_metallic(),
    "mesh_1280_light": lambda: scenes.mesh_scene(3, area_light=True),
    "mesh_5120": lambda: scenes.mesh_scene(4),
}


@pytest.mark.parametrize("name", list(SCENES))
def test_bvh_intersect_matches_reference_traversal(name):
    cam_args, objs, heur = SCENES[name]()
    scene = Scene(objs, 1e-6, 1e6, heur, HDRI, device=0)
    osc = _oracle.OracleScene(objs, 1e-6, 1e6, heur, HDRI)
    o, d = _rays(3000, 17)
    t = np.zeros(len(o))
    obj = np.zeros(len(o), dtype=np.int64)
    _ffi.check(scene._L.rayrs_test_intersect(scene._h, o.ctypes.data, d.ctypes.data, len(o), 0, t.ctypes.data,
                                             obj.ctypes.data), "rayrs_test_intersect")
    # against the reference's recursive, un-narrowed traversal (bvh.rs:391-415)
    rt, robj = osc.intersect_many(o, d, 1e-6, 1e6, traversal=0)
    assert (robj >= 0).sum() > 100
    if name == "five_with_single_leaf":
        assert (scene.export_bvh()[1] >> 30 == 2).any()       # the reference's tree has a direct leaf ...
        kinds = scene.export_wide()[1] >> 30
        assert (kinds == 1).all() or (kinds == 3).any()        # ... which the walk tree holds as a gated one-primitive range
    assert np.array_equal(obj, robj)
    assert np.array_equal(bits(t), bits(rt))


def test_bvh_intersect_degenerate_rays_on_an_integer_grid():
    """Origins on box planes, zero (and negative-zero) direction components, everything on an
    integer grid: the slab test's inf/NaN cases (geometry.rs:458-513) as the rule, not the
    exception.  GPU traversal over the walk tree against the reference's recursion."""
    from rayrs_amd.api import BvhHeuristic, Emission, Object
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
    d[::7, 1] = -0.0
    o, d = np.ascontiguousarray(o), np.ascontiguousarray(d)
    for heur in (BvhHeuristic.Sah(1000), BvhHeuristic.Midpoint):
        scene = Scene(objs, 1e-6, 1e6, heur, HDRI, device=0)
        osc = _oracle.OracleScene(objs, 1e-6, 1e6, heur, HDRI)
        t = np.zeros(n)
        obj = np.zeros(n, dtype=np.int64)
        _ffi.check(scene._L.rayrs_test_intersect(scene._h, o.ctypes.data, d.ctypes.data, n, 0, t.ctypes.data,
                                                 obj.ctypes.data), "rayrs_test_intersect")
        rt, robj = osc.intersect_many(o, d, 1e-6, 1e6, traversal=0)
        assert (robj >= 0).sum() > 500
        assert np.array_equal(obj, robj)
        assert np.array_equal(bits(t), bits(rt))


def _gpu_intersect(scene, o, d, exact):
    t = np.zeros(len(o))
    obj = np.zeros(len(o), dtype=np.int64)
    _ffi.check(scene._L.rayrs_test_intersect(scene._h, o.ctypes.data, d.ctypes.data, len(o), int(exact), t.ctypes.data,
                                             obj.ctypes.data), "rayrs_test_intersect")
    return t, obj


def test_the_default_walk_returns_the_reference_hit_where_the_fast_walk_loses_it():
    """The default walk / rayrs_test_intersect(exact = 1): the walk of the gate tree (the
    reference's groups behind their gating boxes) culls nothing (cull margin +infinity), so it tests exactly the
    primitives BvhTree::intersect tests and returns the reference's closest hit BY
    CONSTRUCTION -- also for the pinned rays of tests/test_walk_tree.py: the one within 1e-9 rad of a triangle's plane, whose
    hit the fast walk's margin of 2^-10 loses (the fast walk returns what the oracle's culled walk returns: the
    neighbour 2 % behind), and for grazing rays of ANY angle, which the fast walk only gets right from 1e-7 rad up."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import fuzz_traversal as F
    hdri = np.zeros((2, 2, 3), dtype=np.float32)
    objs, heur, scale, verts, idx = F.scene_for(79)
    t0, t1 = 1e-6 * scale, 1e9 * scale
    scene = Scene(objs, t0, t1, heur, hdri, device=0)
    osc = _oracle.OracleScene(objs, t0, t1, heur, hdri).use_walk_tree(scene)
    o = np.array([[0.8461539702186601, -0.3178647511202013, 1.6666324107517303]])
    d = np.array([[-4.878144810174007e-10, -0.00017608737629874798, -0.004121392011531156]])
    rt, robj = osc.intersect_batch(o, d, t0, t1, traversal=0)     # the reference's recursion
    wt, wobj = osc.intersect_batch(o, d, t0, t1, traversal=2)     # the walk with the default margin
    assert robj[0] >= 0 and wobj[0] != robj[0]
    t, obj = _gpu_intersect(scene, o, d, exact=0)
    assert obj[0] == wobj[0] and bits(t)[0] == bits(wt)[0]       # the heuristic's known failure, reproduced
    t, obj = _gpu_intersect(scene, o, d, exact=1)
    assert obj[0] == robj[0] and bits(t)[0] == bits(rt)[0]       # no culling: the reference's answer
    # the other bet: a ray along a triangle's plane from 130 000 scene sizes away, on which the reference's own test
    # accepts a neighbour the ray passes beside by more than the 1/64 the default tree's leaf boxes allow
    objs, heur, scale, verts, idx = F.scene_for(2)
    t0, t1 = 1e-6 * scale, 1e9 * scale
    scene = Scene(objs, t0, t1, heur, hdri, device=0)
    osc = _oracle.OracleScene(objs, t0, t1, heur, hdri).use_walk_tree(scene)
    o = np.array([[-2.8981278659447747, -633.8240400572021, 259665.5017321564]])
    d = np.array([[0.0002220828721502402, 0.05051414085108857, -20.691401286965466]])
    rt, robj = osc.intersect_batch(o, d, t0, t1, traversal=0)
    wt, wobj = osc.intersect_batch(o, d, t0, t1, traversal=2)
    assert robj[0] >= 0 and wobj[0] >= 0 and wobj[0] != robj[0]
    t, obj = _gpu_intersect(scene, o, d, exact=0)
    assert obj[0] == wobj[0] and bits(t)[0] == bits(wt)[0]
    t, obj = _gpu_intersect(scene, o, d, exact=1)
    assert obj[0] == robj[0] and bits(t)[0] == bits(rt)[0]
    # rays aimed along triangles' own planes at every angle down to 1e-12 rad, and general rays, on sliver meshes and
    # nearly flat sheets: exact = 1 equals the recursion on all of them
    n_hits = 0
    for seed in (1, 2, 3, 5, 79):
        objs, heur, scale, verts, idx = F.scene_for(seed)
        t0, t1 = 1e-6 * scale, 1e9 * scale
        scene = Scene(objs, t0, t1, heur, hdri, device=0)
        osc = _oracle.OracleScene(objs, t0, t1, heur, hdri)
        rr = np.random.default_rng(seed * 104729 + 5)
        og, dg = F.rays_for(rr, verts, scale, 20_000)
        oz, dz, eps = F.grazing_rays(rr, verts, idx, scale, 60_000)
        assert (eps < 1e-9).any() and (eps > 1e-5).any()
        o, d = np.ascontiguousarray(np.vstack([og, oz])), np.ascontiguousarray(np.vstack([dg, dz]))
        rt, robj = osc.intersect_batch(o, d, t0, t1, traversal=0)
        t, obj = _gpu_intersect(scene, o, d, exact=1)
        assert np.array_equal(obj, robj) and np.array_equal(bits(t), bits(rt)), seed
        n_hits += int((robj >= 0).sum())
    assert n_hits > 50_000


MATERIALS = {
    "lambertian": Material.LambertianDiffuse((0.8, 0.7, 0.6)),
    "reflect": Material.Reflect((0.8, 0.8, 0.8)),
    "refract": Material.Refract((1, 1, 1), 1.45),
    "glass": Material.Glass((0.8, 0.8, 0.8), 1.45),
    "ct_metal_rough": Material.CookTorrance((1, 1, 1), 0.5, Fresnel.SchlickMetallic((0.8, 0.8, 0.8))),
    "ct_metal_smooth": Material.CookTorrance((1, 1, 1), 0.01, Fresnel.SchlickMetallic((0.722, 0.451, 0.2))),
    "ct_dielectric": Material.CookTorrance((0.9, 0.9, 0.9), 0.2, Fresnel.SchlickDielectric(1.45)),
    "ct_refract": Material.CookTorranceRefract((1, 1, 1), 0.09, 1.45),
    "ct_glass_smooth": Material.CookTorranceGlass((1, 1, 1), 0.01, 1.45),
    "ct_glass_rough": Material.CookTorranceGlass((1, 1, 1), 0.25, 1.45),
    "plastic": Material.Plastic((0.8, 0.8, 0.8), (1, 1, 1), 0.05, 1.45),
    "no_reflect": Material.NoReflect(),
}


def _unit(v):
    return v / np.sqrt((v * v).sum(axis=1, keepdims=True))


@pytest.mark.parametrize("name", list(MATERIALS))
def test_material_evaluate_bit_exact(name):
    mat = MATERIALS[name]
    r = np.random.default_rng(11)
    n = 4000
    normal = _unit(r.normal(size=(n, 3)))
    view = _unit(r.normal(size=(n, 3)))
    view[:50] = normal[:50]                       # normal incidence
    view[50:100] = _unit(view[50:100] - normal[50:100] * (view[50:100] * normal[50:100]).sum(1, keepdims=True))  # grazing
    normal[100:110] = [0.0, 1.0, 0.0]             # the floor's normal
    normal = np.ascontiguousarray(normal)
    view = np.ascontiguousarray(view)
    key = r.integers(0, 2 ** 63, n, dtype=np.uint64)
    sc = np.zeros(n, dtype=np.int32)
    col = np.zeros((n, 3))
    dr = np.zeros((n, 3))
    nd = np.zeros(n, dtype=np.uint32)
    m = mat.desc()
    _ffi.check(_ffi.lib().rayrs_test_material(0, C.byref(m), normal.ctypes.data, view.ctypes.data, key.ctypes.data,
                                              n, sc.ctypes.data, col.ctypes.data, dr.ctypes.data, nd.ctypes.data),
               "rayrs_test_material")
    rsc, rcol, rdr, rnd = _oracle.material_evaluate(mat, normal, view, key)
    assert np.array_equal(sc, rsc)
    assert np.array_equal(nd, rnd)
    hit = rsc == 1
    assert np.array_equal(bits(col[hit]), bits(rcol[hit]))
    assert np.array_equal(bits(dr[hit]), bits(rdr[hit]))


def test_background_bit_exact():
    cam_args, objs, heur = scenes.diffuse_single_sphere()
    scene = Scene(objs, 1e-6, 1e6, heur, HDRI, device=0)
    osc = _oracle.OracleScene(objs, 1e-6, 1e6, heur, HDRI)
    r = np.random.default_rng(5)
    d = r.normal(size=(5000, 3))
    # integral texel coordinates (black), poles, the phi seam
    d[:6] = [[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]]
    d[6] = [-1.0, 0.0, -1e-300]
    d = np.ascontiguousarray(d)
    out = np.zeros_like(d)
    _ffi.check(scene._L.rayrs_test_background(scene._h, d.ctypes.data, len(d), out.ctypes.data),
               "rayrs_test_background")
    assert np.array_equal(bits(out), bits(osc.background(d)))
