"""The reference's scene functions (rayrs-lib/src/test_scenes.rs) restated as
plain descriptions, plus the mesh scenes the benchmark configurations name.

Every function returns (camera_args, objects, heuristic): `camera_args` are the
seven Camera::new arguments, `objects` a list of rayrs_amd.api.Object.  The same
description feeds the HIP library (api.Scene) and, in tests, the CPU oracle.
Resolutions: the reference derives pixels from (width cm, height cm, ppi)
(lib.rs:113, :153-177); `camera_for_resolution` picks ppi = 100 (ppc 254) and
width = W/254 cm, height = H/254 cm so that x_pixels = W and y_pixels = H.
"""
import os

from . import procedural
from .api import Axis, BvhHeuristic, Emission, Fresnel, Material, Object

SAH_1000 = BvhHeuristic.Sah(1000)  # test_scenes.rs:40


def _floor():  # test_scenes.rs:15-21
    mat = Material.CookTorrance((1.0, 1.0, 1.0), 0.5, Fresnel.SchlickMetallic((0.8, 0.8, 0.8)))
    return Object.plane(Axis.Y, -25.0, 25.0, -25.0, 25.0, 0.0, mat, Emission.Dark())


def camera_for_resolution(camera_args, width_px: int, height_px: int):
    origin, up, lookat, fov, _, _, _ = camera_args
    return (origin, up, lookat, fov, width_px / 254.0, height_px / 254.0, 100)


# ---- single sphere family, test_scenes.rs:14-68

_SINGLE_CAM = ((0.0, 5.0, 10.0), (0.0, 1.0, 0.0), (0.0, 1.0, 0.0), 50.0, 1920.0 / 500.0, 1080.0 / 500.0, 100)


def single_sphere(mat):
    sphere = Object.sphere(1.0, (0.0, 1.0, 0.0), mat, Emission.Dark())
    return _SINGLE_CAM, [_floor(), sphere], SAH_1000


def copper_single_sphere():  # :46-53
    return single_sphere(Material.CookTorrance((1, 1, 1), 0.05, Fresnel.SchlickMetallic((0.722, 0.451, 0.2))))


def glass_single_sphere():  # :55-58
    return single_sphere(Material.Glass((0.8, 0.8, 0.8), 1.45))


def diffuse_single_sphere():  # :60-63
    return single_sphere(Material.LambertianDiffuse((0.8, 0.8, 0.8)))


def cook_torrance_glass_single_sphere():  # :65-68
    return single_sphere(Material.CookTorranceGlass((0.8, 0.8, 0.8), 0.05, 1.45))


# ---- sphere rows, test_scenes.rs:169-274

_ROW_CAM = ((0.0, 10.0, 20.0), (0.0, 1.0, 0.0), (0.0, 1.0, 0.0), 72.0, 1920.0 / 500.0, 400.0 / 500.0, 125)


def multiple_spheres(mats, cam=_ROW_CAM):  # :169-211
    n = len(mats)
    objs = [_floor()]
    for i, m in enumerate(mats):
        # Vec3::unit_y() + Vec3::new(2.2 * (i - len/2) as f64, 0, 0)
        objs.append(Object.sphere(1.0, (0.0 + 2.2 * float(i - n // 2), 1.0 + 0.0, 0.0 + 0.0), m, Emission.Dark()))
    return cam, objs, SAH_1000


def cook_torrance_spheres_metallic():  # :213-224
    return multiple_spheres([
        Material.CookTorrance((1, 1, 1), 0.01 * float(4 * i + 1), Fresnel.SchlickMetallic((0.8, 0.8, 0.8)))
        for i in range(7)])


def cook_torrance_spheres_plastic():  # :226-239
    return multiple_spheres([Material.Plastic((0.8, 0.8, 0.8), (1, 1, 1), 0.01 * float(4 * i + 1), 1.45)
                             for i in range(7)])


def cook_torrance_spheres_frosted_glass():  # :241-256
    return multiple_spheres([Material.CookTorranceGlass((1, 1, 1), 0.01 * float(4 * i + 1), 1.45)
                             for i in range(7)])


def cook_torrance_spheres_cook_torrance_refract():  # :258-274
    mats = [Material.CookTorranceRefract((1, 1, 1), 0.01 * float(4 * i + 1), 1.45) for i in range(6)]
    mats.insert(0, Material.Refract((1, 1, 1), 1.45))
    return multiple_spheres(mats)


def material_test():  # :276-331
    mats = [
        Material.LambertianDiffuse((0.8, 0.8, 0.8)),
        Material.Plastic((0.8, 0.8, 0.8), (1, 1, 1), 0.05, 1.45),
        Material.Reflect((0.8, 0.8, 0.8)),
        Material.CookTorrance((1, 1, 1), 0.05, Fresnel.SchlickMetallic((0.8, 0.8, 0.8))),
        Material.Glass((1, 1, 1), 1.45),
        Material.CookTorranceGlass((1, 1, 1), 0.05, 1.45),
        Material.NoReflect(),
    ]
    cam = ((0.0, 3.0, 20.0), (0.0, 1.0, 0.0), (0.0, 1.0, 0.0), 90.0, 1920.0 / 500.0, 250.0 / 500.0, 125)
    return multiple_spheres(mats, cam)


# ---- mesh scenes (obj_scene, test_scenes.rs:70-109, with a generated mesh)

def write_mesh_ply(level: int, ply_path):
    """The mesh of mesh_scene(level) as a binary PLY file (written to a temporary name and renamed: a reader never
    sees half a file)."""
    from . import io
    verts, idx = procedural.blob_mesh(level)
    tmp = f"{ply_path}.{os.getpid()}.tmp"
    io.save_ply(tmp, verts, idx, binary=True)
    os.replace(tmp, ply_path)


def mesh_scene(level: int, mat=None, area_light: bool = False, ply_path=None, ply_exists: bool = False):
    """Floor + one closed mesh of 20 * 4**level triangles (+ an emissive
    rectangle).  The light is a Lambertian plane with Emission::Emissive: an
    emitter whose material does not scatter contributes nothing (lib.rs:550).
    With ply_path the mesh is written to that PLY file and read back through the
    library's PLY loader, which is how a real scanned model would arrive
    (ply_exists: somebody has written it -- write_mesh_ply -- and it is only read:
    the ranks of a node share one file)."""
    if mat is None:
        mat = Material.CookTorrance((1, 1, 1), 0.05, Fresnel.SchlickMetallic((0.722, 0.451, 0.2)))  # copper_suzanne
    if ply_path is not None:
        from . import io
        if not ply_exists:
            write_mesh_ply(level, ply_path)
        verts, idx = io.load_ply(ply_path)
    else:
        verts, idx = procedural.blob_mesh(level)
    objs = [_floor()]
    objs += Object.from_triangles(verts, idx, mat, Emission.Dark())
    if area_light:
        objs.append(Object.plane(Axis.YRev, -1.5, 1.5, -1.5, 1.5, 4.5, Material.LambertianDiffuse((0.8, 0.8, 0.8)),
                                 Emission.new(6.0, (1.0, 0.95, 0.9))))
    return _SINGLE_CAM, objs, SAH_1000


# A second framing of the mesh scenes for bench.py's secondary line: the reference's obj_scene camera
# (test_scenes.rs:91-99, _SINGLE_CAM above) sees mostly floor; from here the mesh fills the frame.
MESH_CLOSE_CAM = ((0.0, 2.1, 3.3), (0.0, 1.0, 0.0), (0.0, 1.2, 0.0), 40.0, 1920.0 / 500.0, 1920.0 / 500.0, 100)


# ---- the benchmark configurations of BASELINE.json

CONFIG_MESH_LEVEL = {3: 6, 5: 8}


def write_config_ply(n: int, ply_path):
    """The PLY file config(n, ply_path, ply_exists=True) reads (configs 3 and 5)."""
    write_mesh_ply(CONFIG_MESH_LEVEL[n], ply_path)


def config(n: int, ply_path=None, ply_exists: bool = False):
    """(camera_args, objects, heuristic, spp, max_bounces) of configs[n-1]."""
    if n == 1:  # single diffuse sphere, 256x256, 64 spp
        cam, objs, h = diffuse_single_sphere()
        return camera_for_resolution(cam, 256, 256), objs, h, 64, 50
    if n == 2:  # Cook-Torrance metallic sphere row, 1024x1024, 256 spp
        cam, objs, h = cook_torrance_spheres_metallic()
        return camera_for_resolution(cam, 1024, 1024), objs, h, 256, 50
    if n == 3:  # ~70k-triangle mesh + area light, 1024x1024, 512 spp
        cam, objs, h = mesh_scene(6, Material.LambertianDiffuse((0.8, 0.8, 0.8)), area_light=True, ply_path=ply_path, ply_exists=ply_exists)
        return camera_for_resolution(cam, 1024, 1024), objs, h, 512, 50
    if n == 4:  # frosted-glass spheres, max depth 32, 2048x2048, 4096 spp
        cam, objs, h = cook_torrance_spheres_frosted_glass()
        return camera_for_resolution(cam, 2048, 2048), objs, h, 4096, 32
    if n == 5:  # 1M-triangle mesh (20 * 4**8 = 1,310,720), 2048x2048, 1024 spp
        cam, objs, h = mesh_scene(8, ply_path=ply_path, ply_exists=ply_exists)
        return camera_for_resolution(cam, 2048, 2048), objs, h, 1024, 50
    raise ValueError("config 1..5")
