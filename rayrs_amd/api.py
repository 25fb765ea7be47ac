"""Host-side mirror of the rayrs-lib interface for the hot path.

Same names, argument order and error behaviour as the reference's Rust API
(paths relative to /root/reference/rayrs-lib/src):

    Material / Fresnel / Emission   material.rs:57-68, :127-131, :1056-1075
    Object::{sphere, plane, triangle, from_triangles, from_spheres, box_geom}   lib.rs:321-506
    Scene::new                      lib.rs:227
    Camera::new / x_pixels / y_pixels   lib.rs:99, :153, :175
    render(...)                     the block loop of rayrs/src/main.rs:57-101

Objects, materials and emissions are plain Python descriptions; nothing is
computed here.  Scene() hands them to librayrs_hip.so through the C ABI
(include/rayrs_hip.h), which builds the BVH and uploads it; render() runs the
gfx950 kernel.  A reference `assert!` becomes a ValueError.
"""
import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _ffi

Vec = Tuple[float, float, float]


class Axis:  # geometry.rs:161-168
    X, XRev, Y, YRev, Z, ZRev = range(6)


class BvhHeuristic:  # bvh.rs:187-191
    Midpoint = ("midpoint", 0)

    @staticmethod
    def Sah(splits: int):
        return ("sah", int(splits))


MAT_LAMBERTIAN, MAT_REFLECT, MAT_REFRACT, MAT_GLASS, MAT_COOK_TORRANCE, MAT_COOK_TORRANCE_REFRACT, \
    MAT_COOK_TORRANCE_GLASS, MAT_PLASTIC, MAT_NO_REFLECT = range(9)


def _v(x) -> Vec:
    a = tuple(float(c) for c in x)
    if len(a) != 3:
        raise ValueError("expected 3 components")
    return a


@dataclass(frozen=True)
class Fresnel:  # material.rs:127-131
    metallic: bool
    ior: float = 0.0
    r0: Vec = (0.0, 0.0, 0.0)

    @staticmethod
    def SchlickDielectric(ior: float) -> "Fresnel":
        return Fresnel(False, float(ior))

    @staticmethod
    def SchlickMetallic(r0) -> "Fresnel":
        return Fresnel(True, 0.0, _v(r0))


@dataclass(frozen=True)
class Material:  # material.rs:57-68
    kind: int
    color: Vec = (0.0, 0.0, 0.0)
    spec_color: Vec = (0.0, 0.0, 0.0)
    alpha: float = 0.0
    ior: float = 0.0
    metallic: bool = False
    r0: Vec = (0.0, 0.0, 0.0)

    # constructors, in the reference's argument order
    @staticmethod
    def LambertianDiffuse(color) -> "Material":  # material.rs:608
        return Material(MAT_LAMBERTIAN, _v(color))

    @staticmethod
    def Reflect(color) -> "Material":  # :629
        return Material(MAT_REFLECT, _v(color))

    @staticmethod
    def Refract(color, ior) -> "Material":  # :650
        return Material(MAT_REFRACT, _v(color), ior=float(ior))

    @staticmethod
    def Glass(color, ior) -> "Material":  # :673
        return Material(MAT_GLASS, _v(color), ior=float(ior))

    @staticmethod
    def CookTorrance(color, alpha, fresnel: Fresnel) -> "Material":  # :705
        return Material(MAT_COOK_TORRANCE, _v(color), alpha=float(alpha), ior=fresnel.ior,
                        metallic=fresnel.metallic, r0=fresnel.r0)

    @staticmethod
    def CookTorranceRefract(color, alpha, ior) -> "Material":  # :832
        return Material(MAT_COOK_TORRANCE_REFRACT, _v(color), alpha=float(alpha), ior=float(ior))

    @staticmethod
    def CookTorranceGlass(color, alpha, ior) -> "Material":  # :863
        return Material(MAT_COOK_TORRANCE_GLASS, _v(color), alpha=float(alpha), ior=float(ior))

    @staticmethod
    def Plastic(color, spec_color, alpha, ior) -> "Material":  # :887
        return Material(MAT_PLASTIC, _v(color), _v(spec_color), float(alpha), float(ior))

    @staticmethod
    def NoReflect() -> "Material":
        return Material(MAT_NO_REFLECT)

    def desc(self) -> _ffi.MaterialDesc:
        d = _ffi.MaterialDesc()
        d.kind = self.kind
        d.metallic = 1 if self.metallic else 0
        d.color[:] = self.color
        d.spec_color[:] = self.spec_color
        d.alpha = self.alpha
        d.ior = self.ior
        d.r0[:] = self.r0
        return d


@dataclass(frozen=True)
class Emission:  # material.rs:1056-1075
    emissive: bool = False
    strength: float = 0.0
    color: Vec = (0.0, 0.0, 0.0)

    @staticmethod
    def Dark() -> "Emission":
        return Emission()

    @staticmethod
    def new(strength, color) -> "Emission":  # Emission::new, :1067
        return Emission(True, float(strength), _v(color))

    Emissive = new

    def desc(self) -> _ffi.EmissionDesc:
        d = _ffi.EmissionDesc()
        d.emissive = 1 if self.emissive else 0
        d.strength = self.strength
        d.color[:] = self.color
        return d


@dataclass
class Object:  # lib.rs:302-306
    """One scene object, or (kind == "mesh") the Vec<Object> that
    Object::from_triangles returns for an indexed triangle mesh."""
    kind: str
    mat: Material
    emission: Emission
    radius: float = 0.0
    origin: Vec = (0.0, 0.0, 0.0)
    axis: int = 0
    umin: float = 0.0
    umax: float = 0.0
    vmin: float = 0.0
    vmax: float = 0.0
    pos: float = 0.0
    p: Tuple[Vec, Vec, Vec] = ((0, 0, 0),) * 3
    verts: Optional[np.ndarray] = field(default=None, repr=False)  # (n,3) f32 or f64
    idx: Optional[np.ndarray] = field(default=None, repr=False)    # (m,3) u32

    @staticmethod
    def sphere(radius, origin, mat, emission) -> "Object":  # lib.rs:321
        return Object("sphere", mat, emission, radius=float(radius), origin=_v(origin))

    @staticmethod
    def plane(axis, umin, umax, vmin, vmax, pos, mat, emission) -> "Object":  # lib.rs:342
        return Object("plane", mat, emission, axis=int(axis), umin=float(umin), umax=float(umax),
                      vmin=float(vmin), vmax=float(vmax), pos=float(pos))

    @staticmethod
    def triangle(p1, p2, p3, mat, emission) -> "Object":  # lib.rs:380
        return Object("triangle", mat, emission, p=(_v(p1), _v(p2), _v(p3)))

    @staticmethod
    def from_triangles(verts, idx, mat, emission) -> List["Object"]:  # lib.rs:407
        verts = np.ascontiguousarray(verts)
        if verts.dtype not in (np.float32, np.float64):
            verts = verts.astype(np.float64)
        idx = np.ascontiguousarray(idx, dtype=np.uint32)
        if verts.ndim != 2 or verts.shape[1] != 3 or idx.ndim != 2 or idx.shape[1] != 3:
            raise ValueError("verts must be (n,3) and idx (m,3)")
        return [Object("mesh", mat, emission, verts=verts, idx=idx)]

    @staticmethod
    def from_spheres(radius, centers, mat, emission) -> List["Object"]:  # lib.rs:422
        return [Object.sphere(radius, c, mat, emission) for c in np.asarray(centers, dtype=np.float64)]

    @staticmethod
    def box_geom(lower_left, upper_right, mat, emission) -> List["Object"]:  # lib.rs:438-506
        ll, ur = _v(lower_left), _v(upper_right)
        return [
            Object.plane(Axis.X, ll[1], ur[1], ll[2], ur[2], ll[0], mat, emission),
            Object.plane(Axis.XRev, ll[1], ur[1], ll[2], ur[2], ur[0], mat, emission),
            Object.plane(Axis.ZRev, ll[0], ur[0], ll[1], ur[1], ll[2], mat, emission),
            Object.plane(Axis.Z, ll[0], ur[0], ll[1], ur[1], ur[2], mat, emission),
            Object.plane(Axis.YRev, ll[0], ur[0], ll[2], ur[2], ll[1], mat, emission),
            Object.plane(Axis.Y, ll[0], ur[0], ll[2], ur[2], ll[1], mat, emission),
        ]


def _d3(v):
    return (C.c_double * 3)(*v)


def flatten_objects(objects) -> List[Object]:
    out = []
    for o in objects:
        if isinstance(o, (list, tuple)):
            out.extend(flatten_objects(o))
        else:
            out.append(o)
    return out


class Camera:
    """Pinhole camera, lib.rs:54-211."""

    def __init__(self, origin, up, lookat, fov, width, height, ppi):
        L = _ffi.lib()
        self.args = (_v(origin), _v(up), _v(lookat), float(fov), float(width), float(height), int(ppi))
        self.desc = _ffi.CameraDesc()
        st = L.rayrs_camera_new(_d3(self.args[0]), _d3(self.args[1]), _d3(self.args[2]), self.args[3],
                                self.args[4], self.args[5], self.args[6], C.byref(self.desc))
        if st == -1:
            raise ValueError("Camera::new: invalid argument (lib.rs:108-111)")
        _ffi.check(st, "rayrs_camera_new")

    def x_pixels(self) -> int:  # lib.rs:153
        return int(self.desc.x_pixels)

    def y_pixels(self) -> int:  # lib.rs:175
        return int(self.desc.y_pixels)


class Scene:
    """Scene::new(objects, z_near, z_far, heuristic, hdri), lib.rs:227.

    hdri: (H, W, 3) float32 array (what image::hdr::HdrDecoder yields at
    main.rs:36-41); it is clipped to [0, 3] inside the library as main.rs:43
    does.  device = -1 builds a host-only scene (BVH inspection, no render).
    """

    def __init__(self, objects, z_near, z_far, heuristic, hdri, device: int = 0):
        L = _ffi.lib()
        self._L = L
        self._h = None
        objs = C.c_void_p()
        _ffi.check(L.rayrs_objects_create(C.byref(objs)), "rayrs_objects_create")
        try:
            for o in flatten_objects(objects):
                m, e = o.mat.desc(), o.emission.desc()
                if o.kind == "sphere":
                    st = L.rayrs_object_sphere(objs, o.radius, _d3(o.origin), C.byref(m), C.byref(e))
                elif o.kind == "plane":
                    st = L.rayrs_object_plane(objs, o.axis, o.umin, o.umax, o.vmin, o.vmax, o.pos, C.byref(m),
                                              C.byref(e))
                elif o.kind == "triangle":
                    st = L.rayrs_object_triangle(objs, _d3(o.p[0]), _d3(o.p[1]), _d3(o.p[2]), C.byref(m),
                                                 C.byref(e))
                elif o.kind == "mesh":
                    fn = (L.rayrs_object_from_triangles_f32 if o.verts.dtype == np.float32
                          else L.rayrs_object_from_triangles_f64)
                    st = fn(objs, o.verts.ctypes.data, o.verts.shape[0], o.idx.ctypes.data, o.idx.shape[0],
                            C.byref(m), C.byref(e))
                else:
                    raise ValueError(f"unknown object kind {o.kind}")
                if st == -1:
                    raise ValueError(f"Object::{o.kind}: invalid argument (a reference assert! would fire)")
                _ffi.check(st, f"rayrs_object_{o.kind}")
            hdri = np.ascontiguousarray(hdri, dtype=np.float32)
            if hdri.ndim != 3 or hdri.shape[2] != 3:
                raise ValueError("hdri must be (H, W, 3)")
            kind, splits = heuristic
            h = C.c_void_p()
            st = L.rayrs_scene_new(objs, float(z_near), float(z_far), 1 if kind == "sah" else 0, int(splits),
                                   hdri.shape[1], hdri.shape[0], hdri.ctypes.data, int(device), C.byref(h))
            if st == -1:
                raise ValueError("Scene::new: invalid argument (lib.rs:234-235, bvh.rs:229)")
            _ffi.check(st, "rayrs_scene_new")
            self._h = h
            self.device = int(device)
        finally:
            L.rayrs_objects_destroy(objs)

    @classmethod
    def _from_handle(cls, L, h, device):
        self = cls.__new__(cls)
        self._L, self._h, self.device = L, h, int(device)
        return self

    def clone_to_device(self, device: int) -> "Scene":
        """The same scene uploaded to another HIP device without building the BVH again
        (one per GPU for render_multi)."""
        h = C.c_void_p()
        _ffi.check(self._L.rayrs_scene_clone_to_device(self._h, int(device), C.byref(h)), "rayrs_scene_clone_to_device")
        return Scene._from_handle(self._L, h, device)

    def set_tuning(self, **kw):
        """The two scheduling choices of include/rayrs_hip.h rayrs_tuning (pool_slots, local_pool); 0 = default.
        They do not change the arithmetic (the routes differ in which primitives a query tests: include/rayrs_hip.h)."""
        t = _ffi.Tuning()
        for k, v in kw.items():
            if k not in dict(_ffi.Tuning._fields_):
                raise ValueError(f"unknown tuning field {k}")
            setattr(t, k, int(v))
        _ffi.check(self._L.rayrs_scene_set_tuning(self._h, C.byref(t)), "rayrs_scene_set_tuning")

    def lab_set(self, **kw):
        """The kernels' development knobs (rayrs_amd/csrc/rayrs_lab.h): for tests and scripts/ubench only."""
        t = _ffi.LabTuning()
        for k, v in kw.items():
            if k not in dict(_ffi.LabTuning._fields_):
                raise ValueError(f"unknown lab field {k}")
            setattr(t, k, int(v))
        _ffi.check(self._L.rayrs_lab_set(self._h, C.byref(t)), "rayrs_lab_set")

    def info(self) -> dict:
        i = _ffi.SceneInfo()
        _ffi.check(self._L.rayrs_scene_info(self._h, C.byref(i)), "rayrs_scene_info")
        d = {n: getattr(i, n) for n, _ in i._fields_ if n not in ("root_box", "hot_box")}
        d["root_box"] = list(i.root_box)
        d["hot_box"] = list(i.hot_box)
        return d

    def export_bvh(self):
        i = self.info()
        box = np.zeros((max(i["n_interior"], 1), 2, 6), dtype=np.float64)
        ref = np.zeros((max(i["n_interior"], 1), 2), dtype=np.uint32)
        prim = np.zeros(max(i["n_prims"], 1), dtype=np.uint32)
        _ffi.check(self._L.rayrs_scene_export_bvh(self._h, box.ctypes.data, ref.ctypes.data, prim.ctypes.data),
                   "rayrs_scene_export_bvh")
        return box[:i["n_interior"]], ref[:i["n_interior"]], prim[:i["n_prims"]]

    def export_wide(self):
        """The four-slot records the fast walk reads: (box[n_wide,4,6], ref[n_wide,4])."""
        i = self.info()
        box = np.zeros((max(i["n_wide"], 1), 4, 6), dtype=np.float64)
        ref = np.zeros((max(i["n_wide"], 1), 4), dtype=np.uint32)
        _ffi.check(self._L.rayrs_scene_export_wide(self._h, box.ctypes.data, ref.ctypes.data),
                   "rayrs_scene_export_wide")
        return box[:i["n_wide"]], ref[:i["n_wide"]]

    def export_gate_tree(self):
        """The records the default walk reads (the reference's leaf groups behind their gating boxes):
        (box[gate_n_wide,4,6], ref[gate_n_wide,4])."""
        i = self.info()
        box = np.zeros((max(i["gate_n_wide"], 1), 4, 6), dtype=np.float64)
        ref = np.zeros((max(i["gate_n_wide"], 1), 4), dtype=np.uint32)
        _ffi.check(self._L.rayrs_scene_export_gate_tree(self._h, box.ctypes.data, ref.ctypes.data),
                   "rayrs_scene_export_gate_tree")
        return box[:i["gate_n_wide"]], ref[:i["gate_n_wide"]]

    def export_hot_tree(self):
        """The gate tree without the scene's hot group (info()["hot_count"] > 0): what the default walk reads there:
        (box[hot_n_wide,4,6], ref[hot_n_wide,4])."""
        i = self.info()
        box = np.zeros((max(i["hot_n_wide"], 1), 4, 6), dtype=np.float64)
        ref = np.zeros((max(i["hot_n_wide"], 1), 4), dtype=np.uint32)
        _ffi.check(self._L.rayrs_scene_export_hot_tree(self._h, box.ctypes.data, ref.ctypes.data),
                   "rayrs_scene_export_hot_tree")
        return box[:i["hot_n_wide"]], ref[:i["hot_n_wide"]]

    def close(self):
        if self._h is not None:
            self._L.rayrs_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def frame_sample_chunk(width: int, height: int, spp: int, requested: int = 4) -> int:
    """The sample chunk bench.py, the CLI and the full-size tests render a frame with: decided
    from the whole frame (never from the number of GPUs sharing it), so every rank count sums a
    pixel's samples in the same order.  0 = one sequential sum (the reference's order)."""
    return int(_ffi.lib().rayrs_frame_sample_chunk(int(width), int(height), int(spp), int(requested)))


def make_params(spp, max_bounces=50, seed=0x5EED, sample_chunk=0, tile_rank=0, tile_ranks=1, out_f64=False,
                count_work=False, exact_traversal=True, fast_traversal=None) -> _ffi.RenderParams:
    """exact_traversal=True (the default): the reference's visit set by construction; False, or fast_traversal=True:
    the fast walk (include/rayrs_hip.h rayrs_render_params.fast_traversal)."""
    p = _ffi.RenderParams()
    p.spp, p.max_bounces, p.seed = int(spp), int(max_bounces), int(seed)
    p.sample_chunk, p.tile_rank, p.tile_ranks = int(sample_chunk), int(tile_rank), int(tile_ranks)
    p.out_format = 1 if out_f64 else 0
    p.count_work = 1 if count_work else 0
    fast = (not exact_traversal) if fast_traversal is None else bool(fast_traversal)
    p.fast_traversal = 1 if fast else 0
    return p


def render(scene: Scene, camera: Camera, spp: int, max_bounces: int = 50, seed: int = 0x5EED, sample_chunk: int = 0,
           tile_rank: int = 0, tile_ranks: int = 1, out_f64: bool = False, count_work: bool = False, out=None,
           exact_traversal: bool = True, fast_traversal=None):
    """The block loop of rayrs/src/main.rs:57-101 on the GPU.

    Returns (image, stats): image is (y_pixels, x_pixels, 3), f32 (what
    Image::pixels_f32 yields, image.rs:224) or f64 with out_f64; stats holds the
    ray/path counters, the NaN/negative pixel counts of main.rs:81-87 and the
    HIP-event time of the kernel.
    """
    L = scene._L
    H, W = camera.y_pixels(), camera.x_pixels()
    dt = np.float64 if out_f64 else np.float32
    if out is None:
        out = np.zeros((H, W, 3), dtype=dt)
    assert out.shape == (H, W, 3) and out.dtype == dt and out.flags.c_contiguous
    p = make_params(spp, max_bounces, seed, sample_chunk, tile_rank, tile_ranks, out_f64, count_work, exact_traversal,
                    fast_traversal)
    st = _ffi.RenderStats()
    _ffi.check(L.rayrs_render(scene._h, C.byref(camera.desc), C.byref(p), out.ctypes.data, C.byref(st)),
               "rayrs_render")
    return out, st.as_dict()


def render_multi(scene_list, camera: Camera, spp: int, max_bounces: int = 50, seed: int = 0x5EED,
                 sample_chunk: int = 0, out_f64: bool = False):
    """The block loop of rayrs/src/main.rs:57-101 over several GPUs inside the library: scene i
    renders the tiles t % n == i on its own host thread and stream, one RCCL reduce of the
    framebuffers assembles the frame on the first scene's device.  Returns (image, summed stats)."""
    L = scene_list[0]._L
    H, W = camera.y_pixels(), camera.x_pixels()
    out = np.zeros((H, W, 3), dtype=np.float64 if out_f64 else np.float32)
    p = make_params(spp, max_bounces, seed, sample_chunk, 0, 1, out_f64, False)
    st = _ffi.RenderStats()
    handles = (C.c_void_p * len(scene_list))(*[s._h for s in scene_list])
    _ffi.check(L.rayrs_render_multi(handles, len(scene_list), C.byref(camera.desc), C.byref(p), out.ctypes.data,
                                    C.byref(st)), "rayrs_render_multi")
    return out, st.as_dict()


def render_launch(scene: Scene, camera: Camera, params: _ffi.RenderParams, out_device_ptr: int, stream: int = 0):
    """Enqueue a render writing a DEVICE buffer (e.g. torch tensor .data_ptr())."""
    _ffi.check(scene._L.rayrs_render_launch(scene._h, C.byref(camera.desc), C.byref(params),
                                            C.c_void_p(out_device_ptr), C.c_void_p(stream)), "rayrs_render_launch")


def render_finish(scene: Scene) -> dict:
    st = _ffi.RenderStats()
    _ffi.check(scene._L.rayrs_render_finish(scene._h, C.byref(st)), "rayrs_render_finish")
    return st.as_dict()
