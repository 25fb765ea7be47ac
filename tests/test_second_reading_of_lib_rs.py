"""A second, independent reading of rayrs-lib/src/lib.rs -- Camera::new / generate_primary_ray (:99-133, :202-210),
Scene::background (:254-285), radiance (:521-560) -- with Sphere / Plane (geometry.rs:106-136, :229-282) and
Emission::emit (material.rs:1077-1084), against the oracle.  Same idea as test_second_reading_of_material_rs.py, whose
Material::evaluate it reuses: plain scalar Python written from the Rust text, the closest hit by brute force over the
objects (what BvhTree::intersect returns whenever no gating box is degenerate: smallest accepted t), tolerance 1e-9.
A path whose second reading skips a decision on the edge (a Russian-roulette draw within 1e-9 of p, ...) is left out."""
import math

import numpy as np

import _oracle
import test_second_reading_of_material_rs as M2
from rayrs_amd import procedural
from rayrs_amd.api import Axis, BvhHeuristic, Emission, Fresnel, Material, Object

HDRI = procedural.make_hdri(96, 48)
T0, T1 = 1e-6, 1e6


def camera(origin, up, lookat, fov, width, height, ppi):  # lib.rs:99-133
    ppc = round(ppi * 2.54)
    z = M2.unit(M2.sub(lookat, origin))
    x = M2.unit(M2.cross(up, z))
    y = M2.unit(M2.cross(z, x))
    return dict(origin=origin, e_x=x, e_y=y, z=M2.scale(z, width / math.tan(math.radians(fov) / 2.0)), ppc=ppc,
                width=width, height=height)


def primary_ray(cam, i, j, rnd):  # lib.rs:202-210
    x = (j + rnd()) / cam["ppc"] - cam["width"] / 2.0
    y = (i + rnd()) / cam["ppc"] - cam["height"] / 2.0
    return cam["origin"], M2.add(M2.add(cam["z"], M2.scale(cam["e_x"], x)), M2.scale(cam["e_y"], y))


def background(hdri, d):  # lib.rs:254-285; the texels clipped to [0, 3] as main.rs:43 does
    H, W = hdri.shape[:2]
    d = M2.unit(d)
    phi = math.atan2(d[2], d[0]) + math.pi
    theta = math.acos(d[1])
    x = phi / (2.0 * math.pi) * (W - 1)
    y = theta / math.pi * (H - 1)
    x_f, x_c, y_f, y_c = math.floor(x), math.ceil(x), math.floor(y), math.ceil(y)
    i, j = int(y_f), int(x_f)
    px = lambda a, b: tuple(float(min(max(c, 0.0), 3.0)) for c in hdri[min(a, H - 1), min(b, W - 1)])
    f = [px(i, j), px(i + 1, j), px(i, j + 1), px(i + 1, j + 1)]
    w = [(x_c - x) * (y_c - y), (x_c - x) * (y - y_f), (x - x_f) * (y_c - y), (x - x_f) * (y - y_f)]
    out = (0.0, 0.0, 0.0)
    for fk, wk in zip(f, w):
        out = M2.add(out, M2.scale(fk, wk))
    return out


def intersect(o, d, ob):  # geometry.rs:106-132 (sphere), :229-271 (rectangle)
    if ob.kind == "sphere":
        od = M2.sub(o, ob.origin)
        a, b, c = M2.dot(d, d), 2.0 * M2.dot(d, od), M2.dot(od, od) - ob.radius * ob.radius
        desc = b * b - 4.0 * a * c
        if desc > 0.0:
            t1, t2 = (-b - math.sqrt(desc)) / (2.0 * a), (-b + math.sqrt(desc)) / (2.0 * a)
            # a ray that starts ON the sphere: t1 is a rounding error around zero, and its SIGN decides whether the far
            # side is found (t1 < 0: t2) or lost (0 <= t1 < tmin: Some(t1), rejected by the leaf).
            # The two readings' origins differ in their last bits (math.radians / tan in the camera): leave such rays out
            if abs(t1) < 1e-9 * max(1.0, abs(t2)) and t2 > T0:
                raise M2.Skip()
            if t1 < 0.0:
                return None if t2 < 0.0 else t2
            return t1
        return None
    ax = ob.axis >> 1
    if d[ax] == 0.0:
        return None
    t = (ob.pos - o[ax]) / d[ax]
    p = M2.add(o, M2.scale(d, t))
    u, v = {0: (p[1], p[2]), 1: (p[0], p[2]), 2: (p[0], p[1])}[ax]
    return t if (ob.umin <= u < ob.umax and ob.vmin <= v < ob.vmax) else None


def normal_at(ob, p):  # geometry.rs:134-136, :273-282
    if ob.kind == "sphere":
        return M2.unit(M2.sub(p, ob.origin))
    n = [0.0, 0.0, 0.0]
    n[ob.axis >> 1] = -1.0 if ob.axis & 1 else 1.0
    return tuple(n)


def radiance(objs, hdri, o, d, max_bounces, rnd):  # lib.rs:521-560
    thr, light = (1.0, 1.0, 1.0), (0.0, 0.0, 0.0)
    rays = 0
    for _ in range(max_bounces):
        rays += 1
        best = None
        for ob in objs:
            t = intersect(o, d, ob)
            if t is not None and T0 < t < T1 and (best is None or t < best[0]):
                if best is not None:
                    M2.near(t, best[0])
                best = (t, ob)
        if best is None:
            return M2.add(light, M2.mulv(thr, background(hdri, d))), rays
        t, ob = best
        p = M2.add(o, M2.scale(d, t))
        n = normal_at(ob, p)
        view = M2.unit(M2.scale(d, -1.0))
        sub_rnd = M2.Draws(rnd.key)
        sub_rnd.n = rnd.n
        out, _ = evaluate_with(ob.mat, n, view, sub_rnd)
        rnd.n = sub_rnd.n
        if out is None:
            return light, rays
        color, newdir = out
        e = ob.emission
        light = M2.add(light, M2.mulv(thr, M2.scale(e.color, e.strength) if e.emissive else (0.0, 0.0, 0.0)))
        thr = M2.mulv(thr, color)
        pmax = max(max(thr[0], thr[1]), thr[2])
        x = rnd()
        M2.near(x, pmax)
        if x > pmax:
            return light, rays
        thr = (thr[0] / pmax, thr[1] / pmax, thr[2] / pmax)
        o, d = p, newdir
    return light, rays


def evaluate_with(mat, n, v, rnd):
    """M2.evaluate with a draw counter that continues the path's (the key is the path's, lib.rs:539 draws from it too)"""
    real = M2.Draws
    try:
        M2.Draws = lambda key: rnd  # evaluate() makes its own counter from the key: hand it the path's instead
        return M2.evaluate(mat, n, v, rnd.key)
    finally:
        M2.Draws = real


def test_paths_of_a_sphere_scene_land_on_the_second_reading():
    floor = Object.plane(Axis.Y, -25.0, 25.0, -25.0, 25.0, 0.0,
                         Material.CookTorrance((1, 1, 1), 0.5, Fresnel.SchlickMetallic((0.8, 0.8, 0.8))), Emission.Dark())
    objs = [floor,
            Object.sphere(1.0, (0.0, 1.0, 0.0), Material.LambertianDiffuse((0.8, 0.8, 0.8)), Emission.Dark()),
            Object.sphere(0.7, (2.2, 0.7, 0.5), Material.Glass((0.9, 0.9, 0.9), 1.45), Emission.new(0.8, (1.0, 0.6, 0.3))),
            Object.sphere(0.6, (-2.0, 0.6, 1.0), Material.Plastic((0.7, 0.2, 0.2), (1, 1, 1), 0.1, 1.45), Emission.Dark()),
            Object.plane(Axis.ZRev, -3.0, 3.0, 0.0, 2.5, -2.5, Material.Reflect((0.9, 0.9, 0.9)), Emission.Dark())]
    cam_args = ((0.0, 4.0, 9.0), (0.0, 1.0, 0.0), (0.0, 1.0, 0.0), 50.0, 0.64, 0.36, 100)
    osc = _oracle.OracleScene(objs, T0, T1, BvhHeuristic.Sah(1000), HDRI)
    ocam = _oracle.OracleCamera(*cam_args)
    cam = camera(*cam_args)
    assert (ocam.x_pixels(), ocam.y_pixels()) == (round(0.64 * cam["ppc"]), round(0.36 * cam["ppc"]))
    r = np.random.default_rng(3)
    checked = long_paths = 0
    for k in range(800):
        i, j = int(r.integers(1, ocam.y_pixels())), int(r.integers(1, ocam.x_pixels()))
        key = int(r.integers(0, 2 ** 63))
        try:
            rnd = M2.Draws(key)
            o, d = primary_ray(cam, i, j, rnd)
            oo, od, draw = ocam.primary_ray(i, j, key)
            assert np.allclose(oo, o, rtol=1e-12, atol=0) and np.allclose(od, d, rtol=1e-12, atol=1e-15)
            want, rays = radiance(objs, HDRI, o, d, 50, rnd)
        except (M2.Skip, ValueError, ZeroDivisionError, OverflowError):
            continue
        got, orays, odraw = osc.radiance(oo, od, 50, key, draw=draw, traversal=0)
        assert orays == rays and odraw == rnd.n, (k, orays, rays)
        assert np.allclose(got, want, rtol=1e-8, atol=1e-12), (k, got, want)
        checked += 1
        long_paths += rays >= 3
    assert checked > 600 and long_paths > 50


def test_background_lands_on_the_second_reading():
    objs = [Object.sphere(1.0, (0.0, 1.0, 0.0), Material.NoReflect(), Emission.Dark())]
    osc = _oracle.OracleScene(objs, T0, T1, BvhHeuristic.Midpoint, HDRI)
    r = np.random.default_rng(5)
    d = r.normal(size=(500, 3)) * 10.0 ** r.uniform(-3, 3, size=(500, 1))
    got = osc.background(d)
    for k in range(len(d)):
        assert np.allclose(got[k], background(HDRI, tuple(d[k])), rtol=1e-9, atol=1e-12), k
