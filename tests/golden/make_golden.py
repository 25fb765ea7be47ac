"""Generates the committed golden vectors of tests/golden/ from the CPU oracle.

The reference itself cannot run here (Rust, no rustc) and is non-deterministic
(OS-seeded RNG), so these vectors are outputs of the C restatement
(oracle/rayrs_oracle.c, portable math, counter RNG), committed so that any later
change of the oracle, the numeric contract or the kernels shows up as a diff.

    python tests/golden/make_golden.py        # rewrites golden_v1.npz and rng_known_answers.txt
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import _oracle  # noqa: E402
from rayrs_amd import procedural, scenes  # noqa: E402
from rayrs_amd.api import Fresnel, Material  # noqa: E402

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

SCENES = {
    "diffuse_single_sphere": scenes.diffuse_single_sphere,
    "spheres_metallic": scenes.cook_torrance_spheres_metallic,
    "spheres_frosted_glass": scenes.cook_torrance_spheres_frosted_glass,
    "material_test": scenes.material_test,
    "mesh_1280_light": lambda: scenes.mesh_scene(3, Material.LambertianDiffuse((0.8, 0.8, 0.8)), area_light=True),
}

FRAME = dict(w=40, h=24, spp=6, max_bounces=50, seed=0x5EED)
TRACE = dict(paths=64, cap=64)  # per golden scene: this many seeded samples of the golden frame, traced bounce by bounce


def trace_samples(name):
    """The (row, col) pixels and sample indices of a scene's traced paths: seeded, spread over the golden frame."""
    r = np.random.default_rng(abs(hash_name(name)) % (2 ** 32))
    rows = r.integers(0, FRAME["h"], TRACE["paths"])
    cols = r.integers(0, FRAME["w"], TRACE["paths"])
    samples = r.integers(0, FRAME["spp"], TRACE["paths"])
    return np.stack([rows, cols], axis=1).astype(np.uint32), samples.astype(np.uint32)


def hash_name(name):
    h = 1469598103934665603
    for c in name.encode():
        h = ((h ^ c) * 1099511628211) % (2 ** 64)
    return h
HDRI_SHAPE = (64, 32)


def unit(v):
    return v / np.sqrt((v * v).sum(axis=1, keepdims=True))


def rays(n, seed):
    r = np.random.default_rng(seed)
    o = r.uniform(-6, 6, (n, 3))
    o[:, 1] = np.abs(o[:, 1]) + 0.05
    target = r.uniform(-2.5, 2.5, (n, 3))
    target[:, 1] = np.abs(target[:, 1])
    return np.ascontiguousarray(o), np.ascontiguousarray(target - o)


def material_inputs(n, seed):
    r = np.random.default_rng(seed)
    normal = unit(r.normal(size=(n, 3)))
    view = unit(r.normal(size=(n, 3)))
    view[:8] = normal[:8]
    key = r.integers(0, 2 ** 63, n, dtype=np.uint64)
    return np.ascontiguousarray(normal), np.ascontiguousarray(view), key


def main():
    _oracle.set_math_mode(False)
    out = {}
    hdri = procedural.make_hdri(*HDRI_SHAPE)
    out["hdri"] = hdri

    # RNG known answers
    r = np.random.default_rng(123)
    rows = [(0, 0, 0, 0), (0x5EED, 0, 0, 0), (0x5EED, 1234567, 1023, 17), (2 ** 64 - 1, 2 ** 22 - 1, 4095, 199)]
    for _ in range(28):
        rows.append((int(r.integers(0, 2 ** 63)), int(r.integers(0, 2 ** 22)), int(r.integers(0, 4096)),
                     int(r.integers(0, 200))))
    with open(os.path.join(HERE, "rng_known_answers.txt"), "w") as f:
        f.write("# seed(hex) pixel sample draw -> 64 random bits (hex); include/rayrs_numeric.h rr_draw_bits(rr_path_key())\n")
        for seed, pixel, sample, draw in rows:
            f.write(f"{seed:x} {pixel} {sample} {draw} {_oracle.rng_bits(seed, pixel, sample, draw):016x}\n")

    # per-function vectors
    for name, mat in MATERIALS.items():
        n, v, k = material_inputs(192, 7)
        sc, col, dr, nd = _oracle.material_evaluate(mat, n, v, k)
        out[f"mat/{name}/scattered"], out[f"mat/{name}/color"] = sc, col
        out[f"mat/{name}/dir"], out[f"mat/{name}/draws"] = dr, nd
    for name, fn in SCENES.items():
        cam_args, objs, heur = fn()
        osc = _oracle.OracleScene(objs, 1e-6, 1e6, heur, hdri, builder=0)  # the literal reference builder
        o, d = rays(256, 11)
        t, obj = osc.intersect_many(o, d, 1e-6, 1e6, traversal=0)
        out[f"isect/{name}/t"], out[f"isect/{name}/obj"] = t, obj
        cam_args = scenes.camera_for_resolution(cam_args, FRAME["w"], FRAME["h"])
        ocam = _oracle.OracleCamera(*cam_args)
        img, st = osc.render(ocam, FRAME["spp"], FRAME["max_bounces"], seed=FRAME["seed"], traversal=0)
        out[f"frame/{name}/rgb"] = img
        out[f"frame/{name}/rays"] = np.array([st["rays"]], dtype=np.uint64)
        # per-path traces: what the query of every bounce found, its t, the throughput, the draw index
        pix, sam = trace_samples(name)
        tr = osc.path_traces(ocam, pix, sam, FRAME["seed"], FRAME["max_bounces"], TRACE["cap"], traversal=0)
        for k, v in tr.items():
            out[f"trace/{name}/{k}"] = v
    cam_args, objs, heur = scenes.diffuse_single_sphere()
    osc = _oracle.OracleScene(objs, 1e-6, 1e6, heur, hdri)
    dirs = np.random.default_rng(5).normal(size=(256, 3))
    dirs[:6] = [[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]]
    out["background/dirs"] = dirs
    out["background/rgb"] = osc.background(dirs)
    np.savez_compressed(os.path.join(HERE, "golden_v1.npz"), **out)
    print("wrote", os.path.join(HERE, "golden_v1.npz"), os.path.getsize(os.path.join(HERE, "golden_v1.npz")), "bytes")


if __name__ == "__main__":
    main()
