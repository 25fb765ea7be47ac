"""A second, independent reading of rayrs-lib/src/material.rs against the oracle.

The oracle (oracle/rayrs_oracle.c) and the device code (rayrs_amd/csrc/device_path.h) are the same functions
by the same hand: that they agree bit for bit cannot reveal a shared misreading of the reference (VERDICT r2,
"parity is twin-transliteration parity above the intersection layer").  The reference holds no numeric vector
for its materials, so nothing can PIN them; what can be done is to read material.rs again, from the top, and
write it down a second time in the plainest possible way -- scalar Python, the expressions copied from the Rust
text, math.* for the elementary functions -- and require the oracle to land on the same numbers.  Tolerance
1e-9 (relative, on colours; absolute on unit directions): platform libm against the build's fdlibm forms
(<= 2 ulp), and Python's operator order where the Rust text leaves it open.  A branch decided by `random < F`
or `x >= 1` can flip when the two sides differ in the last bits: samples within 1e-9 of a decision are skipped.

Every function cites the lines it was read from.  The random numbers are the build's own contract
(include/rayrs_numeric.h: SplitMix64 finaliser on key + (draw + 1) * golden, 53 bits), restated here too."""
import math

import numpy as np
import pytest

import _oracle
from rayrs_amd.api import Fresnel, Material

M64 = (1 << 64) - 1
GOLDEN = 0x9E3779B97F4A7C15


def mix64(z):
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


class Draws:
    """rand::random::<f64>() in program order: the draw-th uniform of the path with this key."""

    def __init__(self, key):
        self.key, self.n = int(key), 0

    def __call__(self):
        bits = mix64((self.key + (self.n + 1) * GOLDEN) & M64)
        self.n += 1
        return (bits >> 11) * 2.0 ** -53


class Skip(Exception):
    """a decision within 1e-9 of its threshold"""


def near(x, y):
    if abs(x - y) <= 1e-9 * max(1.0, abs(y)):
        raise Skip()


# ---- vecmath.rs
def dot(a, b): return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]
def add(a, b): return (a[0] + b[0], a[1] + b[1], a[2] + b[2])
def sub(a, b): return (a[0] - b[0], a[1] - b[1], a[2] - b[2])
def scale(a, s): return (a[0] * s, a[1] * s, a[2] * s)
def mulv(a, b): return (a[0] * b[0], a[1] * b[1], a[2] * b[2])
def cross(a, b): return (a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0])
def unit(a): return scale(a, 1.0 / math.sqrt(dot(a, a)))
def is_zeros(a): return a[0] == 0.0 and a[1] == 0.0 and a[2] == 0.0


def orthonormal_basis(n):  # vecmath.rs:341-352
    e1 = unit((n[2], 0.0, -n[0])) if abs(n[0]) > abs(n[1]) else unit((0.0, n[2], -n[1]))
    return e1, unit(cross(n, e1))


# ---- material.rs:1472-1518
def schlick_scalar(ior_curr, ior_new, normal, view):
    r0 = (ior_curr - ior_new) / (ior_curr + ior_new)
    r0 = r0 * r0
    return r0 + (1.0 - r0) * (1.0 - dot(normal, view)) ** 5


def schlick_vec(r0, normal, view):
    p = (1.0 - dot(normal, view)) ** 5
    return tuple(r + (1.0 - r) * p for r in r0)


def reflect(normal, view):
    return sub(scale(normal, 2.0 * dot(view, normal)), view)


def refract(normal, view, ior_ratio):
    cos_theta = dot(view, normal)
    sin_theta = math.sqrt(1.0 - cos_theta * cos_theta)
    near(ior_ratio * sin_theta, 1.0)
    if ior_ratio * sin_theta > 1.0:
        return None
    par = scale(sub(scale(normal, cos_theta), view), ior_ratio)
    perp = scale(normal, -math.sqrt(1.0 - dot(par, par)))
    return add(perp, par)


# ---- ScatteringDirection, material.rs:1189-1231
def ior_ratio_of(entering, ior): return 1.0 / ior if entering else ior
def flip(entering, n): return n if entering else scale(n, -1.0)
def iors(entering, ior): return (1.0, ior) if entering else (ior, 1.0)


class CT:  # struct CookTorrance, material.rs:194-200, ctor :705-713 (alpha2 = alpha * alpha)
    def __init__(self, color, alpha, metallic, ior=0.0, r0=(0, 0, 0)):
        self.color, self.alpha2, self.metallic, self.ior, self.r0 = color, alpha * alpha, metallic, ior, r0

    def fresnel(self, normal, view, entering):  # Fresnel::value, :1457-1469
        if self.metallic:
            return schlick_vec(self.r0, normal, view)
        c, n = iors(entering, self.ior)
        f = schlick_scalar(c, n, normal, view)
        return (f, f, f)

    def beckmann(self, tan_theta_h, nh):  # the expression at :940, :1310, :1414
        return math.exp(-tan_theta_h * tan_theta_h / self.alpha2) / (math.pi * self.alpha2 * nh ** 4)

    def brdf(self, normal, light, view):  # :1276-1322
        nv, nl = abs(dot(normal, view)), abs(dot(normal, light))
        h = add(view, light)
        if nv == 0.0 or nl == 0.0 or is_zeros(h):
            return (0.0, 0.0, 0.0)
        h = unit(h)
        nh = dot(normal, h)
        tan_theta_h = math.tan(math.acos(nh))
        if math.isinf(tan_theta_h):
            return (0.0, 0.0, 0.0)
        hv = dot(h, view)
        g = min(2.0 * nh * nv / hv, min(2.0 * nh * nl / hv, 1.0))
        f = self.fresnel(h, view, True)
        return scale(scale(scale(mulv(self.color, f), self.beckmann(tan_theta_h, nh)), g), 1.0 / (4.0 * nv * nl))

    def btdf(self, normal, light, view, entering):  # :1362-1442
        nv, nl = abs(dot(normal, view)), abs(dot(normal, light))
        ior_ratio = ior_ratio_of(entering, self.ior)
        h = add(light, scale(view, ior_ratio)) if ior_ratio > 1.0 else sub(scale(view, -ior_ratio), light)
        if nv == 0.0 or nl == 0.0 or is_zeros(h):
            return (0.0, 0.0, 0.0)
        h = unit(h)
        nh = dot(normal, h)
        tan_theta_h = math.tan(math.acos(nh))
        if math.isinf(tan_theta_h):
            return (0.0, 0.0, 0.0)
        hl, hv = abs(dot(h, light)), abs(dot(h, view))
        g = min(2.0 * nh * nv / hv, min(2.0 * nh * nl / hv, 1.0))
        denom = (ior_ratio * hv + hl) ** 2
        norm_fac = hv * hl / (nv * nl)
        f = self.fresnel(h, view, entering)
        c = mulv(self.color, tuple(1.0 - x for x in f))
        return scale(c, self.beckmann(tan_theta_h, nh) * g * norm_fac * ior_ratio * ior_ratio / denom)

    def pdf_value(self, normal, light, view):  # Pdf::Beckmann(_, Reflect).value, :915-941
        h = add(light, view)
        if is_zeros(h):
            return 1.0
        h = unit(h)
        nh = abs(dot(normal, h))
        tan_theta_h = math.tan(math.acos(nh))
        if math.isinf(tan_theta_h):
            return 1.0
        return self.beckmann(tan_theta_h, nh)

    def generate(self, normal, rnd, with_value):  # Pdf::Beckmann.generate :1006-1020, MicrofacetDistribution :1139-1161
        e1, e2 = orthonormal_basis(normal)
        phi = 2.0 * math.pi * rnd()
        tan2theta = -self.alpha2 * math.log(1.0 - rnd())
        costheta = 1.0 / math.sqrt(1.0 + tan2theta)
        sintheta = math.sqrt(1.0 - costheta * costheta)
        x, y = math.cos(phi) * sintheta, math.sin(phi) * sintheta
        h = add(add(scale(e1, x), scale(e2, y)), scale(normal, costheta))
        if not with_value:
            return h, None
        nh = dot(normal, h)
        return h, math.exp(-tan2theta / self.alpha2) / (math.pi * self.alpha2 * nh ** 4)

    def evaluate_reflection(self, normal, h, view, light, pdf):  # :721-758
        near(dot(h, view), 0.0)
        if dot(h, view) < 0.0:
            return None
        nl = dot(normal, light)
        near(nl, 0.0)
        if nl < 0.0:
            return None
        color = scale(scale(scale(self.brdf(normal, light, view), nl), 1.0 / pdf), 4.0 * dot(h, light))
        return None if is_zeros(color) else (color, light)

    def evaluate_refraction(self, normal, h, view, light, pdf, entering, ior_ratio):  # :764-812
        near(dot(h, view), 0.0)
        if dot(h, view) < 0.0:
            return None
        nl = dot(normal, light)
        near(nl, 0.0)
        if nl > 0.0:
            return None
        hl, hv = abs(dot(h, light)), abs(dot(h, view))
        dwh_dwi = hl / (ior_ratio * hv + hl) ** 2
        color = scale(scale(self.btdf(normal, light, view, entering), abs(nl)), 1.0 / (ior_ratio * ior_ratio))
        color = scale(color, 1.0 / (pdf * dwh_dwi))
        return None if is_zeros(color) else (color, light)

    def scatter(self, normal, view, rnd):  # impl Bsdf for CookTorrance, :403-424
        h, _ = self.generate(normal, rnd, False)
        light = reflect(h, view)
        return self.evaluate_reflection(normal, h, view, light, self.pdf_value(normal, light, view))


def lambertian(color, normal, rnd):  # :259-281 with Pdf::Cosine :913, :982-993 and brdf :1233-1243
    e1, e2 = orthonormal_basis(normal)
    u = rnd()
    phi = 2.0 * math.pi * rnd()
    x, y, z = math.cos(phi) * math.sqrt(u), math.sin(phi) * math.sqrt(u), math.sqrt(1.0 - u)
    light = add(add(scale(e1, x), scale(e2, y)), scale(normal, z))
    nl = dot(normal, light)
    return scale(scale(scale(color, 1.0 / math.pi), nl), 1.0 / (nl * (1.0 / math.pi))), light


def evaluate(m: Material, normal, view, key):
    """Material::evaluate, material.rs:91-109 -> None (NoScatter) or (color, light); and the draws consumed."""
    rnd = Draws(key)
    k = m.kind
    if k == 0:
        out = lambertian(m.color, normal, rnd)
    elif k == 1:  # Reflect :283-301; brdf :1254-1265; Dirac pdf value 1
        light = reflect(normal, view)
        out = scale(scale(m.color, 1.0 / abs(dot(normal, light))), dot(normal, light)), light
    elif k == 2:  # Refract :303-337; btdf :1333-1351
        entering = dot(normal, view) > 0.0
        n = flip(entering, normal)
        light = refract(n, view, ior_ratio_of(entering, m.ior))
        if light is None:
            out = None
        else:
            b = (0.0, 0.0, 0.0) if dot(light, view) > 0.0 else scale(m.color, 1.0 / abs(dot(n, light)))
            out = scale(b, abs(dot(n, light))), light
    elif k == 3:  # Glass :339-401
        cos_theta = dot(normal, view)
        entering = cos_theta > 0.0
        n = flip(entering, normal)
        sin2theta = 1.0 - cos_theta * cos_theta
        r = ior_ratio_of(entering, m.ior)
        near(r * r * sin2theta, 1.0)
        do_reflect = r * r * sin2theta >= 1.0
        if not do_reflect:
            c, nw = iors(entering, m.ior)
            f = schlick_scalar(c, nw, n, view)
            x = rnd()
            near(x, f)
            do_reflect = x < f
        if do_reflect:
            light = reflect(n, view)
            out = scale(scale(m.color, 1.0 / abs(dot(n, light))), dot(n, light)), light
        else:
            light = refract(n, view, r)
            b = (0.0, 0.0, 0.0) if dot(light, view) > 0.0 else scale(m.color, 1.0 / abs(dot(n, light)))
            out = scale(b, abs(dot(n, light))), light
    elif k == 4:
        out = CT(m.color, m.alpha, m.metallic, m.ior, m.r0).scatter(normal, view, rnd)
    elif k == 5:  # CookTorranceRefract :426-467
        ct = CT(m.color, m.alpha, False, m.ior)
        entering = dot(normal, view) > 0.0
        n = flip(entering, normal)
        r = ior_ratio_of(entering, m.ior)
        h, value = ct.generate(n, rnd, True)
        h = flip(entering, h)
        light = refract(h, view, r)
        out = None if light is None else ct.evaluate_refraction(n, h, view, light, value, entering, r)
    elif k == 6:  # CookTorranceGlass :469-565
        ct = CT(m.color, m.alpha, False, m.ior)
        h, value = ct.generate(normal, rnd, True)
        entering = dot(normal, view) > 0.0
        h = flip(entering, h)
        n = flip(entering, normal)
        cos_theta = dot(h, view)
        r = ior_ratio_of(entering, m.ior)
        sin2 = 1.0 - cos_theta * cos_theta
        near(r * r * sin2, 1.0)
        if r * r * sin2 >= 1.0:
            out = ct.evaluate_reflection(n, h, view, reflect(h, view), value)
        else:
            c, nw = iors(entering, m.ior)
            f = schlick_scalar(c, nw, h, view)
            x = rnd()
            near(x, f)
            if x < f:
                out = ct.evaluate_reflection(n, h, view, reflect(h, view), value)
                out = None if out is None else (scale(out[0], 1.0 / f), out[1])
            else:
                out = ct.evaluate_refraction(n, h, view, refract(h, view, r), value, entering, r)
                out = None if out is None else (scale(out[0], 1.0 / (1.0 - f)), out[1])
    elif k == 7:  # Plastic :567-593; ctor :887-901 (spec_color on the Cook-Torrance layer, dielectric)
        f = schlick_scalar(1.0, m.ior, normal, view)
        x = rnd()
        near(x, f)
        if x < f:
            out = CT(m.spec_color, m.alpha, False, m.ior).scatter(normal, view, rnd)
            out = None if out is None else (scale(out[0], 1.0 / f), out[1])
        else:
            out = lambertian(m.color, normal, rnd)
    else:
        out = None  # Material::NoReflect, :107
    return out, rnd.n


MATERIALS = [
    ("lambertian", Material.LambertianDiffuse((0.8, 0.5, 0.3)), False),
    ("reflect", Material.Reflect((0.9, 0.8, 0.7)), False),
    ("refract", Material.Refract((0.9, 1.0, 0.8), 1.45), True),
    ("glass", Material.Glass((0.8, 0.8, 0.9), 1.5), True),
    ("cook_torrance_metal", Material.CookTorrance((1.0, 0.9, 0.8), 0.3, Fresnel.SchlickMetallic((0.72, 0.45, 0.2))), False),
    ("cook_torrance_dielectric", Material.CookTorrance((0.9, 0.9, 0.9), 0.12, Fresnel.SchlickDielectric(1.45)), False),
    ("cook_torrance_refract", Material.CookTorranceRefract((1.0, 0.95, 0.9), 0.2, 1.45), True),
    ("cook_torrance_glass", Material.CookTorranceGlass((0.9, 1.0, 1.0), 0.15, 1.45), True),
    ("plastic", Material.Plastic((0.7, 0.2, 0.2), (1.0, 1.0, 1.0), 0.1, 1.45), False),
    ("no_reflect", Material.NoReflect(), False),
]


@pytest.mark.parametrize("name,mat,two_sided", MATERIALS, ids=[m[0] for m in MATERIALS])
def test_oracle_lands_on_the_second_reading(name, mat, two_sided):
    r = np.random.default_rng(17)
    n_s = 1500
    n = r.normal(size=(n_s, 3)); n /= np.linalg.norm(n, axis=1, keepdims=True)
    w = r.normal(size=(n_s, 3)); w /= np.linalg.norm(w, axis=1, keepdims=True)
    v = n + 0.97 * w; v /= np.linalg.norm(v, axis=1, keepdims=True)
    if two_sided:
        v[::2] *= -1.0  # from inside the medium
    t = np.cross(n[:20], w[:20])
    v[:20] = t / np.linalg.norm(t, axis=1, keepdims=True) + 0.0141 * n[:20]  # grazing views
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    keys = r.integers(0, 2 ** 63, size=n_s, dtype=np.uint64)
    sc, col, dr, nd = _oracle.material_evaluate(mat, n, v, keys)
    checked = scattered = 0
    for i in range(n_s):
        try:
            out, draws = evaluate(mat, tuple(n[i]), tuple(v[i]), keys[i])
        except (Skip, ValueError, ZeroDivisionError, OverflowError):
            continue  # a decision on the edge, or math.* raising where IEEE arithmetic returns inf / NaN
        checked += 1
        assert (out is not None) == bool(sc[i]), (name, i)
        assert draws == nd[i], (name, i)
        if out is None:
            continue
        scattered += 1
        color, light = out
        assert np.allclose(col[i], color, rtol=1e-9, atol=1e-300), (name, i, col[i], color)
        assert np.allclose(dr[i], light, rtol=0, atol=1e-9), (name, i)
    assert checked > 0.9 * n_s
    if name != "no_reflect":
        assert scattered > 0.3 * n_s
