"""Analytic properties of the oracle's materials.

The reference holds no numeric test for material.rs, and the oracle and the HIP kernels were
written from the same source text, so bit-equality between them cannot see a shared misreading.
These tests do not come from that text: they check what the physics and the estimator
`weight = f * |n.l| / pdf` demand of ANY correct implementation of material.rs:259-593 --
mirror and Snell geometry, weights that equal the colour where the reference's pdf is exact,
Fresnel-weighted branch frequencies, energy bounds, hemisphere sides -- on large seeded samples,
including grazing directions."""
import math

import numpy as np
import pytest

import _oracle
from rayrs_amd.api import Fresnel, Material

N = 6000


def frames(n, seed, grazing=False):
    """Unit normals, unit views in the upper hemisphere of the normal (or the lower one with flip)."""
    r = np.random.default_rng(seed)
    nrm = r.normal(size=(n, 3))
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    v = r.normal(size=(n, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    cos = (v * nrm).sum(1, keepdims=True)
    v = np.where(cos < 0, v - 2 * cos * nrm, v)        # into the hemisphere of n
    if grazing:                                         # squeeze towards the horizon: n.v in (0, 0.05]
        cos = (v * nrm).sum(1, keepdims=True)
        tang = v - cos * nrm
        tang /= np.linalg.norm(tang, axis=1, keepdims=True)
        c = r.uniform(1e-6, 0.05, (n, 1))
        v = tang * np.sqrt(1 - c * c) + nrm * c
    keys = r.integers(0, 2 ** 63, n, dtype=np.uint64)
    return nrm, v, keys


def dot(a, b):
    return (a * b).sum(1)


def test_lambertian_weight_is_the_colour_and_directions_are_cosine_distributed():
    col = np.array([0.8, 0.5, 0.2])
    n, v, k = frames(N, 1)
    sc, w, l, draws = _oracle.material_evaluate(Material.LambertianDiffuse(col), n, v, k)
    assert sc.all() and (draws == 2).all()
    assert np.allclose(w, col, rtol=1e-14)                       # (c/pi * n.l) / (n.l / pi)
    assert np.allclose(np.linalg.norm(l, axis=1), 1.0, atol=1e-12)
    cos = dot(n, l)
    assert (cos >= -1e-15).all()
    assert abs(cos.mean() - 2.0 / 3.0) < 0.01                    # E[cos] under p = cos/pi
    assert abs((cos ** 2).mean() - 0.5) < 0.01                   # E[cos^2] = 1/2


def test_reflect_is_a_mirror_with_the_colour_as_weight():
    col = np.array([0.9, 0.8, 0.7])
    n, v, k = frames(N, 2)
    sc, w, l, draws = _oracle.material_evaluate(Material.Reflect(col), n, v, k)
    assert sc.all() and (draws == 0).all()
    assert np.allclose(l, 2 * dot(n, v)[:, None] * n - v, atol=1e-14)
    assert np.allclose(dot(n, l), dot(n, v), atol=1e-14)
    assert np.allclose(w, col, rtol=1e-13)


@pytest.mark.parametrize("inside", [False, True])
def test_refract_obeys_snell(inside):
    ior = 1.45
    n, v, k = frames(N, 3)
    if inside:
        v = v - 2 * dot(n, v)[:, None] * n                     # below the surface: n.v < 0
    sc, w, l, draws = _oracle.material_evaluate(Material.Refract((1, 1, 1), ior), n, v, k)
    eta = ior if inside else 1.0 / ior                          # material.rs:1200-1205
    sin_i = np.sqrt(np.maximum(0.0, 1 - dot(n, v) ** 2))
    can = eta * sin_i <= 1.0
    assert np.array_equal(sc.astype(bool), can)                 # total internal reflection -> NoScatter (:1502-1518)
    s = sc.astype(bool)
    sin_t = np.sqrt(np.maximum(0.0, 1 - dot(n[s], l[s]) ** 2))
    assert np.allclose(sin_t, eta * sin_i[s], atol=1e-12)       # Snell
    assert (np.sign(dot(n[s], l[s])) == -np.sign(dot(n[s], v[s]))).all()   # through the surface
    # v, n and l are coplanar
    assert np.allclose(np.einsum("ij,ij->i", np.cross(n[s], v[s]), l[s]), 0.0, atol=1e-12)
    assert np.allclose(w[s], 1.0, rtol=1e-13)
    assert (draws == 0).all()


def test_glass_branches_conserve_and_follow_schlick():
    ior, col = 1.45, np.array([1.0, 1.0, 1.0])
    n, v, k = frames(40000, 4)
    sc, w, l, draws = _oracle.material_evaluate(Material.Glass(col, ior), n, v, k)
    assert sc.all()
    assert np.allclose(w, 1.0, rtol=1e-13)                      # both branches: the Fresnel factor cancels (:339-401)
    reflected = dot(n, l) > 0
    cos = dot(n, v)
    r0 = ((1 - ior) / (1 + ior)) ** 2
    fres = r0 + (1 - r0) * (1 - cos) ** 5
    # frequency of the reflect branch in bins of cos(theta) against Schlick's approximation
    for lo, hi in ((0.0, 0.1), (0.1, 0.3), (0.3, 0.6), (0.6, 1.0)):
        m = (cos >= lo) & (cos < hi)
        assert m.sum() > 1000
        assert abs(reflected[m].mean() - fres[m].mean()) < 4.0 * math.sqrt(0.25 / m.sum()) + 1e-3
    assert (draws == 1).all()
    # from inside, beyond the critical angle: always reflected, no draw
    vin = v - 2 * dot(n, v)[:, None] * n
    sc2, w2, l2, d2 = _oracle.material_evaluate(Material.Glass(col, ior), n, vin, k)
    sin2 = 1 - dot(n, vin) ** 2
    tir = ior * ior * sin2 >= 1.0
    assert tir.sum() > 1000 and (d2[tir] == 0).all()
    assert (dot(n[tir], l2[tir]) < 0).all()                     # stays inside


CT = {
    "metal_rough": Material.CookTorrance((1, 1, 1), 0.5, Fresnel.SchlickMetallic((1.0, 1.0, 1.0))),
    "metal_smooth": Material.CookTorrance((1, 1, 1), 0.05, Fresnel.SchlickMetallic((0.8, 0.8, 0.8))),
    "dielectric": Material.CookTorrance((0.9, 0.9, 0.9), 0.2, Fresnel.SchlickDielectric(1.45)),
    "plastic": Material.Plastic((0.8, 0.8, 0.8), (1, 1, 1), 0.05, 1.45),
    "ct_glass": Material.CookTorranceGlass((1, 1, 1), 0.1, 1.45),
    "ct_refract": Material.CookTorranceRefract((1, 1, 1), 0.1, 1.45),
}


@pytest.mark.parametrize("name", list(CT))
@pytest.mark.parametrize("grazing", [False, True])
def test_microfacet_weights_are_finite_and_non_negative(name, grazing):
    n, v, k = frames(N, 5 + grazing, grazing=grazing)
    sc, w, l, draws = _oracle.material_evaluate(CT[name], n, v, k)
    s = sc.astype(bool)
    assert s.sum() > (10 if grazing else N // 20)
    assert np.isfinite(w[s]).all() and (w[s] >= 0).all()
    assert np.allclose(np.linalg.norm(l[s], axis=1), 1.0, atol=1e-9)
    assert (w[~s] == 0).all()
    if name in ("metal_rough", "metal_smooth", "dielectric"):
        assert (dot(n[s], l[s]) >= 0).all()                     # evaluate_reflection rejects n.l < 0 (:721-758)
        assert (draws == 2).all()


def test_white_metal_reflects_no_more_than_it_receives():
    """A white Cook-Torrance metal (r0 = 1: F = 1) is lossless except for shadowing/masking and rejected
    samples, so the mean weight lies in (0, 1 + noise] and approaches 1 for a smooth surface seen from above."""
    r = np.random.default_rng(9)
    n = np.tile(np.array([0.0, 1.0, 0.0]), (N, 1))
    for alpha, lo in ((0.05, 0.97), (0.5, 0.6)):
        mat = Material.CookTorrance((1, 1, 1), alpha, Fresnel.SchlickMetallic((1.0, 1.0, 1.0)))
        th = math.radians(30.0)
        v = np.tile(np.array([math.sin(th), math.cos(th), 0.0]), (N, 1))
        k = r.integers(0, 2 ** 63, N, dtype=np.uint64)
        sc, w, l, d = _oracle.material_evaluate(mat, n, v, k)
        mean = w[:, 0].mean()                                    # NoScatter counts as zero
        assert lo < mean < 1.03, (alpha, mean)


def test_plastic_mixes_its_layers_by_schlick():
    n, v, k = frames(30000, 11)
    sc, w, l, draws = _oracle.material_evaluate(CT["plastic"], n, v, k)
    cos = dot(n, v)
    r0 = ((1 - 1.45) / (1 + 1.45)) ** 2
    fres = r0 + (1 - r0) * (1 - cos) ** 5
    assert (draws == 3).all()                                    # choice, then two sampling draws (:567-593)
    # the diffuse branch has the exact colour as weight; everything else went through the specular layer
    diffuse = np.isclose(w, 0.8, rtol=1e-13).all(axis=1) & sc.astype(bool)
    assert abs(diffuse.mean() - (1 - fres).mean()) < 0.02
