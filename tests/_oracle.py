"""ctypes wrapper of the CPU oracle (oracle/librayrs_oracle.so).

Test infrastructure: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg only.  Takes the same scene descriptions (rayrs_amd.api.Object
lists) the product takes, so one description drives both sides.
"""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(ROOT, "oracle", "librayrs_oracle.so")


class OrcMaterial(C.Structure):
    _fields_ = [("kind", C.c_int32), ("metallic", C.c_int32), ("color", C.c_double * 3),
                ("spec_color", C.c_double * 3), ("alpha", C.c_double), ("ior", C.c_double),
                ("r0", C.c_double * 3)]


class OrcEmission(C.Structure):
    _fields_ = [("emissive", C.c_int32), ("pad", C.c_int32), ("strength", C.c_double), ("color", C.c_double * 3)]


class OrcCamera(C.Structure):
    _fields_ = [("origin", C.c_double * 3), ("e_x", C.c_double * 3), ("e_y", C.c_double * 3),
                ("z", C.c_double * 3), ("width", C.c_double), ("height", C.c_double), ("ppc", C.c_uint32),
                ("x_pixels", C.c_uint32), ("y_pixels", C.c_uint32)]


class OrcStats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("paths", C.c_uint64), ("nan_pixels", C.c_uint64),
                ("neg_pixels", C.c_uint64), ("interior_visits", C.c_uint64), ("tri_tests", C.c_uint64),
                ("sphere_tests", C.c_uint64), ("plane_tests", C.c_uint64), ("escaped_paths", C.c_uint64),
                ("seconds", C.c_double)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class OrcFlatInfo(C.Structure):
    _fields_ = [("n_interior", C.c_uint32), ("n_prims", C.c_uint32), ("root_ref", C.c_uint32),
                ("depth", C.c_uint32), ("root_box", C.c_double * 6), ("n_wide", C.c_uint32),
                ("wide_root_ref", C.c_uint32), ("wide_depth", C.c_uint32), ("reserved", C.c_uint32)]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    L = C.CDLL(LIB_PATH)
    vp, dp = C.c_void_p, C.POINTER(C.c_double)
    mp, ep = C.POINTER(OrcMaterial), C.POINTER(OrcEmission)
    L.orc_scene_create.restype = vp
    L.orc_scene_destroy.argtypes = [vp]
    L.orc_scene_destroy.restype = None
    L.orc_add_sphere.argtypes = [vp, C.c_double, dp, mp, ep]
    L.orc_add_plane.argtypes = [vp, C.c_int] + [C.c_double] * 5 + [mp, ep]
    L.orc_add_triangle.argtypes = [vp, dp, dp, dp, mp, ep]
    L.orc_add_triangles.argtypes = [vp, vp, C.c_uint32, vp, C.c_uint32, mp, ep]
    L.orc_scene_build.argtypes = [vp, C.c_double, C.c_double, C.c_int, C.c_uint32, C.c_int, C.c_uint32, C.c_uint32,
                                  vp]
    L.orc_camera_new.argtypes = [dp, dp, dp, C.c_double, C.c_double, C.c_double, C.c_uint32, C.POINTER(OrcCamera)]
    L.orc_render.argtypes = [vp, C.POINTER(OrcCamera), C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32,
                             C.c_uint32, C.c_int, C.c_int, vp, C.POINTER(OrcStats)]
    L.orc_math.restype = C.c_double
    L.orc_math.argtypes = [C.c_int, C.c_double, C.c_double]
    L.orc_rng_bits.restype = C.c_uint64
    L.orc_rng_bits.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32]
    L.orc_aabb_intersect.argtypes = [dp, dp, dp, C.c_double, C.c_double]
    L.orc_sphere_intersect.argtypes = [C.c_double, dp, dp, dp, dp]
    L.orc_plane_intersect.argtypes = [C.c_int] + [C.c_double] * 5 + [dp, dp, dp]
    L.orc_triangle_intersect.argtypes = [dp, dp, dp, dp, dp, dp]
    L.orc_triangle_normal.argtypes = [dp, dp, dp, dp]
    L.orc_triangle_normal.restype = None
    L.orc_scene_bbox.argtypes = [vp, dp, dp, dp, dp]
    L.orc_object_boxes.argtypes = [vp, vp]
    L.orc_bvh_intersect.restype = C.c_int64
    L.orc_bvh_intersect.argtypes = [vp, dp, dp, C.c_double, C.c_double, C.c_int, dp]
    L.orc_material_evaluate.argtypes = [mp, dp, dp, dp, C.c_uint64, C.POINTER(C.c_uint32), dp, dp]
    L.orc_background.argtypes = [vp, dp, dp]
    L.orc_background.restype = None
    L.orc_primary_ray.argtypes = [C.POINTER(OrcCamera), C.c_uint32, C.c_uint32, C.c_uint64, C.POINTER(C.c_uint32),
                                  dp, dp]
    L.orc_primary_ray.restype = None
    L.orc_radiance.restype = C.c_uint32
    L.orc_radiance.argtypes = [vp, dp, dp, C.c_uint32, C.c_uint64, C.POINTER(C.c_uint32), C.c_int, dp]
    L.orc_flatten_info.argtypes = [vp, C.POINTER(OrcFlatInfo)]
    L.orc_flatten_export.argtypes = [vp, vp, vp, vp]
    L.orc_flatten_export_wide.argtypes = [vp, vp, vp]
    L.orc_set_wide.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp]
    L.orc_set_hot_group.argtypes = [vp, vp, C.c_uint32, C.c_uint32]
    L.orc_vec_op.argtypes = [C.c_int, dp, dp, C.c_double, C.c_double, dp]
    L.orc_vec_op.restype = None
    L.orc_vec_scalar.argtypes = [C.c_int, dp, dp]
    L.orc_vec_scalar.restype = C.c_double
    L.orc_orthonormal_basis.argtypes = [dp, dp, dp]
    L.orc_orthonormal_basis.restype = None
    L.orc_aabb_expand.argtypes = [dp, dp, dp]
    L.orc_aabb_expand.restype = None
    L.orc_bvh_intersect_batch.argtypes = [vp, C.c_uint64, vp, vp, C.c_double, C.c_double, C.c_int, C.c_int, vp, vp]
    L.orc_cull_margin_probe.argtypes = [vp, C.c_uint64, vp, vp, C.c_double, C.c_double, C.c_int, vp]
    L.orc_path_trace.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, C.c_int, C.c_uint32, vp, vp, vp, vp,
                                 C.POINTER(C.c_double)]
    L.orc_path_trace.restype = C.c_uint32
    L.orc_set_cull_margin.argtypes = [C.c_double]
    L.orc_set_cull_margin.restype = None
    L.orc_set_math_mode.argtypes = [C.c_int]
    L.orc_set_math_mode.restype = None
    _lib = L
    return L


def d3(v):
    return (C.c_double * 3)(*[float(c) for c in v])


def mat_desc(m) -> OrcMaterial:
    d = OrcMaterial()
    d.kind = m.kind
    d.metallic = 1 if m.metallic else 0
    d.color[:] = m.color
    d.spec_color[:] = m.spec_color
    d.alpha = m.alpha
    d.ior = m.ior
    d.r0[:] = m.r0
    return d


def emis_desc(e) -> OrcEmission:
    d = OrcEmission()
    d.emissive = 1 if e.emissive else 0
    d.strength = e.strength
    d.color[:] = e.color
    return d


def set_cull_margin(rel: float):
    """Experiments only: the relative margin of the ordered walks' closest-hit culling (default: the kernel's, 2^-10)."""
    lib().orc_set_cull_margin(float(rel))


def set_math_mode(libm: bool):
    lib().orc_set_math_mode(1 if libm else 0)


class OracleCamera:
    def __init__(self, origin, up, lookat, fov, width, height, ppi):
        self.desc = OrcCamera()
        st = lib().orc_camera_new(d3(origin), d3(up), d3(lookat), float(fov), float(width), float(height), int(ppi),
                                  C.byref(self.desc))
        if st != 0:
            raise ValueError("Camera::new assert")

    def x_pixels(self):
        return int(self.desc.x_pixels)

    def y_pixels(self):
        return int(self.desc.y_pixels)

    def primary_ray(self, i, j, key, draw=0):
        o, d = (C.c_double * 3)(), (C.c_double * 3)()
        dr = C.c_uint32(draw)
        lib().orc_primary_ray(C.byref(self.desc), i, j, key, C.byref(dr), o, d)
        return np.array(o), np.array(d), dr.value


class OracleScene:
    """Scene::new on the oracle.  builder: 0 literal reference algorithm, 1 swept."""

    def __init__(self, objects, z_near, z_far, heuristic, hdri, builder=1):
        from rayrs_amd.api import flatten_objects
        L = lib()
        self._L = L
        self._h = L.orc_scene_create()
        for o in flatten_objects(objects):
            m, e = mat_desc(o.mat), emis_desc(o.emission)
            if o.kind == "sphere":
                st = L.orc_add_sphere(self._h, o.radius, d3(o.origin), C.byref(m), C.byref(e))
            elif o.kind == "plane":
                st = L.orc_add_plane(self._h, o.axis, o.umin, o.umax, o.vmin, o.vmax, o.pos, C.byref(m), C.byref(e))
            elif o.kind == "triangle":
                st = L.orc_add_triangle(self._h, d3(o.p[0]), d3(o.p[1]), d3(o.p[2]), C.byref(m), C.byref(e))
            elif o.kind == "mesh":
                v = np.ascontiguousarray(o.verts, dtype=np.float64)  # f32 -> f64 is exact
                st = L.orc_add_triangles(self._h, v.ctypes.data, v.shape[0], o.idx.ctypes.data, o.idx.shape[0],
                                         C.byref(m), C.byref(e))
            else:
                raise ValueError(o.kind)
            if st != 0:
                raise ValueError(f"oracle rejected object {o.kind}")
        hdri = np.ascontiguousarray(hdri, dtype=np.float32)
        kind, splits = heuristic
        st = L.orc_scene_build(self._h, float(z_near), float(z_far), 1 if kind == "sah" else 0, int(splits),
                               int(builder), hdri.shape[1], hdri.shape[0], hdri.ctypes.data)
        if st != 0:
            raise ValueError("Scene::new assert")

    def close(self):
        if self._h:
            self._L.orc_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def flat_info(self):
        i = OrcFlatInfo()
        assert self._L.orc_flatten_info(self._h, C.byref(i)) == 0
        return {"n_interior": i.n_interior, "n_prims": i.n_prims, "root_ref": i.root_ref, "depth": i.depth,
                "root_box": list(i.root_box), "n_wide": i.n_wide, "wide_root_ref": i.wide_root_ref,
                "wide_depth": i.wide_depth}

    def use_product_walk(self, product_scene, fast=False, hot=None):
        """The walk the product makes: traversal=2 then visits what its traversal kernel visits -- by default the gate
        tree with nothing culled (the reference's visit set; on a scene with a hot group: the tree without that group
        and the group beside it, as the kernels do -- hot=False takes the whole gate tree all the same, which is what
        the local-pool route and rayrs_lab hot_group=0xffffffff walk), with fast=True the tree of single primitives with
        closest-hit culling (rayrs_render_params.fast_traversal)."""
        info = product_scene.info()
        if hot is None:
            hot = not fast and info["hot_count"] > 0 and not info["local_pool"]
        if hot:
            assert not fast and info["hot_count"] > 0
            box, ref = product_scene.export_hot_tree()
            box, ref = np.ascontiguousarray(box), np.ascontiguousarray(ref)
            assert self._L.orc_set_wide(self._h, info["hot_n_wide"], info["hot_root_ref"], info["hot_depth"],
                                        box.ctypes.data, ref.ctypes.data) == 0
            hb = np.array(info["hot_box"], dtype=np.float64)
            assert self._L.orc_set_hot_group(self._h, hb.ctypes.data, info["hot_first"], info["hot_count"]) == 0
        else:
            self.use_walk_tree(product_scene, gate=not fast)
        self._walk_margin = 2.0 ** -10 if fast else float("inf")
        return self

    def _with_margin(self, traversal, call):
        m = getattr(self, "_walk_margin", None)
        if traversal != 2 or m is None:
            return call()
        try:
            set_cull_margin(m)
            return call()
        finally:
            set_cull_margin(2.0 ** -10)

    def use_walk_tree(self, product_scene, gate=False):
        """Take the four-slot records the kernels walk from the product (a rayrs_amd.Scene, host-only
        or on a device): traversal=2 then makes the kernel's walk on the kernel's data.  gate=True: the
        tree the default walk reads (and the local-pool route's gates come from)."""
        info = product_scene.info()
        box, ref = product_scene.export_gate_tree() if gate else product_scene.export_wide()
        box = np.ascontiguousarray(box)
        ref = np.ascontiguousarray(ref)
        pre = "gate_" if gate else "wide_"
        self._walk_margin = None  # (the caller sets the cull margin: _oracle.set_cull_margin)
        assert self._L.orc_set_wide(self._h, info["gate_n_wide" if gate else "n_wide"], info[pre + "root_ref"],
                                    info[pre + "depth"], box.ctypes.data, ref.ctypes.data) == 0
        return self

    def export_wide(self):
        i = self.flat_info()
        box = np.zeros((max(i["n_wide"], 1), 4, 6), dtype=np.float64)
        ref = np.zeros((max(i["n_wide"], 1), 4), dtype=np.uint32)
        assert self._L.orc_flatten_export_wide(self._h, box.ctypes.data, ref.ctypes.data) == 0
        return box[:i["n_wide"]], ref[:i["n_wide"]]

    def export_bvh(self):
        i = self.flat_info()
        box = np.zeros((max(i["n_interior"], 1), 2, 6), dtype=np.float64)
        ref = np.zeros((max(i["n_interior"], 1), 2), dtype=np.uint32)
        prim = np.zeros(max(i["n_prims"], 1), dtype=np.uint32)
        assert self._L.orc_flatten_export(self._h, box.ctypes.data, ref.ctypes.data, prim.ctypes.data) == 0
        return box[:i["n_interior"]], ref[:i["n_interior"]], prim[:i["n_prims"]]

    def render(self, cam: OracleCamera, spp, max_bounces=50, seed=0x5EED, sample_chunk=0, rows=None, nthreads=None,
               traversal=0):
        H, W = cam.y_pixels(), cam.x_pixels()
        out = np.zeros((H, W, 3), dtype=np.float64)
        st = OrcStats()
        r0, r1 = rows if rows is not None else (0, H)
        if nthreads is None:
            nthreads = os.cpu_count() or 1
        rc = self._with_margin(traversal, lambda: self._L.orc_render(
            self._h, C.byref(cam.desc), int(spp), int(max_bounces), int(seed), int(sample_chunk), int(r0), int(r1),
            int(nthreads), int(traversal), out.ctypes.data, C.byref(st)))
        assert rc == 0, "orc_render failed"
        return out, st.as_dict()

    def intersect(self, o, d, tmin, tmax, traversal=0):
        t = C.c_double(0.0)
        obj = self._L.orc_bvh_intersect(self._h, d3(o), d3(d), float(tmin), float(tmax), int(traversal), C.byref(t))
        return int(obj), t.value

    def intersect_many(self, o, d, tmin, tmax, traversal=0):
        n = len(o)
        ts = np.zeros(n)
        objs = np.full(n, -1, dtype=np.int64)
        for i in range(n):
            ob, t = self.intersect(o[i], d[i], tmin, tmax, traversal)
            objs[i] = ob
            ts[i] = t if ob >= 0 else 0.0
        return ts, objs

    def intersect_batch(self, o, d, tmin, tmax, traversal=0, nthreads=None):
        """Bvh::intersect for n rays in one call, on all cores: (t[n], object[n]) with -1 for a miss."""
        o = np.ascontiguousarray(o, dtype=np.float64)
        d = np.ascontiguousarray(d, dtype=np.float64)
        n = len(o)
        ts = np.zeros(n)
        objs = np.full(n, -1, dtype=np.int64)
        rc = self._L.orc_bvh_intersect_batch(self._h, n, o.ctypes.data, d.ctypes.data, float(tmin), float(tmax),
                                             int(traversal), int(nthreads or os.cpu_count() or 1), ts.ctypes.data,
                                             objs.ctypes.data)
        assert rc == 0
        ts[objs < 0] = 0.0
        return ts, objs

    def cull_margin_probe(self, o, d, tmin, tmax, nthreads=None):
        """(largest (box entry - t) / t over accepted hits and the boxes around them, hits in front of a box of
        theirs, hits beyond the kernel's cull margin) on the product's walk tree (use_walk_tree first)."""
        o = np.ascontiguousarray(o, dtype=np.float64)
        d = np.ascontiguousarray(d, dtype=np.float64)
        out = np.zeros(3)
        rc = self._L.orc_cull_margin_probe(self._h, len(o), o.ctypes.data, d.ctypes.data, float(tmin), float(tmax),
                                           int(nthreads or os.cpu_count() or 1), out.ctypes.data)
        assert rc == 0
        return float(out[0]), int(out[1]), int(out[2])

    def background(self, dirs):
        dirs = np.asarray(dirs, dtype=np.float64)
        out = np.zeros_like(dirs)
        rgb = (C.c_double * 3)()
        for i in range(len(dirs)):
            self._L.orc_background(self._h, d3(dirs[i]), rgb)
            out[i] = rgb[:]
        return out

    def path_traces(self, cam, pixels, samples, seed, max_bounces=50, cap=64, traversal=0):
        """Traces of the samples (row, col, sample index) as orc_render runs them: dict of arrays n[k], obj[k, cap],
        t[k, cap], thr[k, cap, 3], draw[k, cap], rgb[k, 3] (entries beyond n[k] are -1 / 0)."""
        k = len(pixels)
        out = dict(n=np.zeros(k, dtype=np.uint32), obj=np.full((k, cap), -1, dtype=np.int64), t=np.zeros((k, cap)),
                   thr=np.zeros((k, cap, 3)), draw=np.zeros((k, cap), dtype=np.uint32), rgb=np.zeros((k, 3)))
        rgb = (C.c_double * 3)()
        for i, ((row, col), s) in enumerate(zip(pixels, samples)):
            obj = np.full(cap, -1, dtype=np.int64); t = np.zeros(cap); thr = np.zeros((cap, 3)); dr = np.zeros(cap, dtype=np.uint32)
            n = self._L.orc_path_trace(self._h, C.byref(cam.desc), int(row), int(col), int(s), int(seed), int(max_bounces),
                                       int(traversal), int(cap), obj.ctypes.data, t.ctypes.data, thr.ctypes.data, dr.ctypes.data, rgb)
            out["n"][i] = n
            out["obj"][i], out["t"][i], out["thr"][i], out["draw"][i], out["rgb"][i] = obj, t, thr, dr, rgb[:]
        return out

    def radiance(self, o, d, max_bounces, key, draw=0, traversal=0):
        rgb = (C.c_double * 3)()
        dr = C.c_uint32(draw)
        n = self._L.orc_radiance(self._h, d3(o), d3(d), int(max_bounces), int(key), C.byref(dr), int(traversal), rgb)
        return np.array(rgb), int(n), dr.value

    def object_boxes(self, n_objects):
        """Object::bbox of every object in insertion order: (n_objects, 6)."""
        out = np.zeros((n_objects, 6), dtype=np.float64)
        assert self._L.orc_object_boxes(self._h, out.ctypes.data) == 0
        return out

    def bbox(self):
        box, cen = (C.c_double * 6)(), (C.c_double * 3)()
        vol, sa = C.c_double(), C.c_double()
        assert self._L.orc_scene_bbox(self._h, box, cen, C.byref(vol), C.byref(sa)) == 0
        return list(box), list(cen), vol.value, sa.value


def material_evaluate(mat, normals, views, keys):
    """Material::evaluate for arrays of (normal, view, key): scattered, color, dir, draws."""
    L = lib()
    m = mat_desc(mat)
    n = len(keys)
    sc = np.zeros(n, dtype=np.int32)
    col = np.zeros((n, 3))
    dr = np.zeros((n, 3))
    nd = np.zeros(n, dtype=np.uint32)
    c, d = (C.c_double * 3)(), (C.c_double * 3)()
    pos = d3((0, 0, 0))
    for i in range(n):
        draw = C.c_uint32(0)
        r = L.orc_material_evaluate(C.byref(m), pos, d3(normals[i]), d3(views[i]), int(keys[i]), C.byref(draw), c, d)
        assert r >= 0
        sc[i] = r
        if r:
            col[i] = c[:]
            dr[i] = d[:]
        nd[i] = draw.value
    return sc, col, dr, nd


def math_fn(fn, x, y=None):
    L = lib()
    x = np.asarray(x, dtype=np.float64)
    y = np.zeros_like(x) if y is None else np.asarray(y, dtype=np.float64)
    return np.array([L.orc_math(fn, float(a), float(b)) for a, b in zip(x, y)])


def rng_bits(seed, pixel, sample, draw):
    return lib().orc_rng_bits(int(seed), int(pixel), int(sample), int(draw))


VEC_OPS = {"add": 0, "sub": 1, "mul": 2, "scale": 3, "cross": 4, "powf": 5, "clip": 6, "unit": 7, "div": 8,
           "div_assign": 9}


def vec_op(name, a, b=(0.0, 0.0, 0.0), s=0.0, t=0.0):
    out = (C.c_double * 3)()
    lib().orc_vec_op(VEC_OPS[name], d3(a), d3(b), float(s), float(t), out)
    return tuple(out)


def vec_dot(a, b):
    return lib().orc_vec_scalar(0, d3(a), d3(b))


def vec_mag2(a):
    return lib().orc_vec_scalar(1, d3(a), d3(a))


def orthonormal_basis(n):
    e1, e2 = (C.c_double * 3)(), (C.c_double * 3)()
    lib().orc_orthonormal_basis(d3(n), e1, e2)
    return tuple(e1), tuple(e2)


def aabb_intersect(box, o, d, tmin, tmax):
    return bool(lib().orc_aabb_intersect((C.c_double * 6)(*box), d3(o), d3(d), tmin, tmax))


def aabb_expand(a, b):
    out = (C.c_double * 6)()
    lib().orc_aabb_expand((C.c_double * 6)(*a), (C.c_double * 6)(*b), out)
    return list(out)
