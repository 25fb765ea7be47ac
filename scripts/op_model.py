"""Driver of scripts/op_model.sh: runs every unit of the shading side alone, N evaluations each at 64/64 lanes, in
a fixed order, so that a rocprofv3 --pmc SQ_INSTS_VALU pass gives the vector instructions ONE evaluation costs
when the unit is compiled alone and every lane works (what bench.py charges as `useful` per unit of shading work).
Units: Material::evaluate per material kind (material.rs:91-109), Scene::background (lib.rs:254-285); `none`
(Material::NoReflect) is the test kernel's own loads and stores, subtracted from the material units."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import rayrs_amd
from rayrs_amd import _ffi, procedural, scenes
from rayrs_amd.api import Fresnel, Material

N = 1 << 21
UNITS = [
    ("none", Material.NoReflect(), False),
    ("lambertian", Material.LambertianDiffuse((0.8, 0.8, 0.8)), False),
    ("reflect", Material.Reflect((0.8, 0.8, 0.8)), False),
    ("refract", Material.Refract((1, 1, 1), 1.45), True),
    ("glass", Material.Glass((0.8, 0.8, 0.8), 1.45), True),
    ("cook_torrance_metal_rough", Material.CookTorrance((1, 1, 1), 0.5, Fresnel.SchlickMetallic((0.8, 0.8, 0.8))), False),
    ("cook_torrance_metal_smooth", Material.CookTorrance((1, 1, 1), 0.05, Fresnel.SchlickMetallic((0.722, 0.451, 0.2))), False),
    ("cook_torrance_dielectric", Material.CookTorrance((1, 1, 1), 0.13, Fresnel.SchlickDielectric(1.45)), False),
    ("cook_torrance_refract", Material.CookTorranceRefract((1, 1, 1), 0.13, 1.45), True),
    ("cook_torrance_glass", Material.CookTorranceGlass((1, 1, 1), 0.13, 1.45), True),
    ("plastic", Material.Plastic((0.8, 0.8, 0.8), (1, 1, 1), 0.13, 1.45), False),
]


def main():
    L = _ffi.lib()
    r = np.random.default_rng(1)
    n = r.normal(size=(N, 3)); n /= np.linalg.norm(n, axis=1, keepdims=True)
    w = r.normal(size=(N, 3)); w /= np.linalg.norm(w, axis=1, keepdims=True)
    v = n + 0.98 * w; v /= np.linalg.norm(v, axis=1, keepdims=True)          # views in the normal's hemisphere
    key = r.integers(0, 2 ** 63, size=N, dtype=np.uint64)
    sc, col, dr, nd = np.zeros(N, np.int32), np.zeros((N, 3)), np.zeros((N, 3)), np.zeros(N, np.uint32)
    order = []
    for name, mat, two_sided in UNITS:
        view = v.copy()
        if two_sided:
            view[::2] *= -1.0                                                 # half of the hits from inside the medium
        m = mat.desc()
        _ffi.check(L.rayrs_test_material(0, C.byref(m), np.ascontiguousarray(n).ctypes.data, np.ascontiguousarray(view).ctypes.data,
                                         key.ctypes.data, N, sc.ctypes.data, col.ctypes.data, dr.ctypes.data, nd.ctypes.data),
                   "rayrs_test_material")
        order.append({"unit": name, "kernel": "test_material_kernel", "n": N, "scattered": float(sc.mean()),
                      "draws": float(nd.mean())})
    cam_args, objs, heur = scenes.diffuse_single_sphere()
    scene = rayrs_amd.Scene(objs, 1e-6, 1e6, heur, procedural.make_hdri(1024, 512), device=0)
    d = np.ascontiguousarray(r.normal(size=(N, 3)))
    out = np.zeros_like(d)
    _ffi.check(L.rayrs_test_background(scene._h, d.ctypes.data, N, out.ctypes.data), "rayrs_test_background")
    order.append({"unit": "background", "kernel": "test_background_kernel", "n": N})
    json.dump(order, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
