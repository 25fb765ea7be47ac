"""ctypes declarations for librayrs_hip.so (include/rayrs_hip.h).

There is no fallback: if the shared library is missing this module raises, so a
product call can never silently run on anything but the HIP build.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RAYRS_HIP_LIB: another build of the same library (same ABI), for the same-box A/B scripts of scripts/ubench/ -- they
# used to copy their alternatives over the tree's library.  Read by this binding, not by the library.
LIB_PATH = os.environ.get("RAYRS_HIP_LIB") or os.path.join(_HERE, "librayrs_hip.so")

# every symbol include/rayrs_hip.h declares
SYMBOLS = [
    "rayrs_strerror", "rayrs_last_error",
    "rayrs_objects_create", "rayrs_objects_destroy", "rayrs_objects_len",
    "rayrs_object_sphere", "rayrs_object_plane", "rayrs_object_triangle",
    "rayrs_object_from_triangles_f32", "rayrs_object_from_triangles_f64",
    "rayrs_object_from_spheres", "rayrs_object_box_geom",
    "rayrs_scene_new", "rayrs_scene_destroy", "rayrs_scene_info", "rayrs_scene_export_bvh",
    "rayrs_scene_export_wide", "rayrs_scene_export_gate_tree", "rayrs_scene_export_hot_tree", "rayrs_scene_clone_to_device", "rayrs_scene_device", "rayrs_scene_set_tuning",
    "rayrs_camera_new",
    "rayrs_frame_sample_chunk", "rayrs_render", "rayrs_render_launch", "rayrs_render_finish", "rayrs_render_multi",
    "rayrs_abi_layout", "rayrs_abi_version",
    "rayrs_io_last_error", "rayrs_buffer_free", "rayrs_ply_load", "rayrs_ply_save", "rayrs_obj_load", "rayrs_obj_load_spheres",
    "rayrs_hdr_load", "rayrs_hdr_save", "rayrs_image_to_bytes", "rayrs_ppm_save", "rayrs_png_save",
]


# rayrs_amd/csrc/rayrs_selftest.h and rayrs_lab.h: private hooks of the library (tests/ and scripts/ only)
PRIVATE_SYMBOLS = ["rayrs_test_math", "rayrs_test_rng", "rayrs_test_intersect", "rayrs_test_path_trace", "rayrs_test_material", "rayrs_test_background",
                   "rayrs_lab_set", "rayrs_lab_round_ms", "rayrs_lab_multi_rehearse"]


class MaterialDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("metallic", C.c_int32), ("color", C.c_double * 3),
                ("spec_color", C.c_double * 3), ("alpha", C.c_double), ("ior", C.c_double),
                ("r0", C.c_double * 3)]


class EmissionDesc(C.Structure):
    _fields_ = [("emissive", C.c_int32), ("pad", C.c_int32), ("strength", C.c_double),
                ("color", C.c_double * 3)]


class CameraDesc(C.Structure):
    _fields_ = [("origin", C.c_double * 3), ("e_x", C.c_double * 3), ("e_y", C.c_double * 3),
                ("z", C.c_double * 3), ("width", C.c_double), ("height", C.c_double),
                ("ppc", C.c_uint32), ("x_pixels", C.c_uint32), ("y_pixels", C.c_uint32)]


class SceneInfo(C.Structure):
    _fields_ = [("n_objects", C.c_uint64), ("n_interior", C.c_uint32), ("n_prims", C.c_uint32),
                ("root_ref", C.c_uint32), ("depth", C.c_uint32), ("compact", C.c_uint32),
                ("n_surfaces", C.c_uint32), ("node_bytes", C.c_uint32), ("prim_bytes", C.c_uint32),
                ("device_bytes", C.c_uint64), ("root_box", C.c_double * 6),
                ("build_seconds", C.c_double), ("n_wide", C.c_uint32), ("wide_root_ref", C.c_uint32),
                ("wide_depth", C.c_uint32), ("local_pool", C.c_uint32), ("gate_n_wide", C.c_uint32),
                ("gate_root_ref", C.c_uint32), ("gate_depth", C.c_uint32),
                ("hot_n_wide", C.c_uint32), ("hot_root_ref", C.c_uint32), ("hot_depth", C.c_uint32),
                ("hot_first", C.c_uint32), ("hot_count", C.c_uint32), ("hot_pad", C.c_uint32),
                ("hot_box", C.c_double * 6)]


class RenderParams(C.Structure):
    _fields_ = [("spp", C.c_uint32), ("max_bounces", C.c_uint32), ("seed", C.c_uint64),
                ("sample_chunk", C.c_uint32), ("tile_rank", C.c_uint32), ("tile_ranks", C.c_uint32),
                ("out_format", C.c_uint32), ("count_work", C.c_uint32), ("fast_traversal", C.c_uint32)]


class RenderStats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("paths", C.c_uint64), ("nan_pixels", C.c_uint64),
                ("neg_pixels", C.c_uint64), ("interior_visits", C.c_uint64), ("tri_tests", C.c_uint64),
                ("sphere_tests", C.c_uint64), ("plane_tests", C.c_uint64), ("escaped_paths", C.c_uint64),
                ("step_wave", C.c_uint64), ("step_lane", C.c_uint64), ("inner_wave", C.c_uint64),
                ("leaf_wave", C.c_uint64), ("interior_ticks", C.c_uint64), ("leaf_ticks", C.c_uint64),
                ("kernel_ms", C.c_double), ("total_ms", C.c_double), ("kernel_launches", C.c_uint64),
                ("trace_ms", C.c_double), ("refill_ticks", C.c_uint64),
                ("surface_hits", C.c_uint64 * 8), ("direct_rays", C.c_uint64), ("hit_ms", C.c_double), ("miss_ms", C.c_double),
                ("local_pool", C.c_uint32), ("exact_walk", C.c_uint32), ("hot_group", C.c_uint32),
                ("stats_pad", C.c_uint32), ("pre_rays", C.c_uint64), ("pre_root_records", C.c_uint64), ("hot_lane", C.c_uint64),
                ("hot_prim_tests", C.c_uint64), ("hot_tri_divided", C.c_uint64)]

    def as_dict(self):
        d = {n: getattr(self, n) for n, _ in self._fields_}
        d["surface_hits"] = list(self.surface_hits)
        return d


class Tuning(C.Structure):
    _fields_ = [("pool_slots", C.c_uint32), ("local_pool", C.c_uint32)]


class LabTuning(C.Structure):
    """rayrs_amd/csrc/rayrs_lab.h: the kernels' development knobs -- tests/ and scripts/ubench/ only, not part of the
    boundary (include/rayrs_hip.h)."""
    _fields_ = [("refill_min", C.c_uint32), ("leaf_min", C.c_uint32), ("static_pct", C.c_uint32),
                ("stack_lds", C.c_uint32), ("hot_records", C.c_uint32), ("trav_blocks_per_cu", C.c_uint32),
                ("eager_light", C.c_uint32), ("local_reserve", C.c_uint32), ("local_segment_items", C.c_uint32),
                ("force_rccl", C.c_uint32), ("gate_tree", C.c_uint32), ("hot_group", C.c_uint32), ("leaf_wait", C.c_uint32), ("flat_blocks_per_cu", C.c_uint32)]


# RAYRS_ABI_VERSION these mirrors were written against: lib() refuses a library of another version
ABI_VERSION = 6

# the order rayrs_abi_layout() reports the public structs in
ABI_STRUCTS = [MaterialDesc, EmissionDesc, CameraDesc, SceneInfo, RenderParams, RenderStats, Tuning]

_lib = None


def lib():
    """Load librayrs_hip.so once; raise loudly when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(make -C rayrs_amd/csrc). There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    L.rayrs_abi_version.argtypes = []
    L.rayrs_abi_version.restype = C.c_uint32
    if L.rayrs_abi_version() != ABI_VERSION:  # (what INTEGRATION.md asks of every binding)
        raise RuntimeError(f"{LIB_PATH} has RAYRS_ABI_VERSION {L.rayrs_abi_version()}, this binding was written against "
                           f"{ABI_VERSION}: rebuild the library (make -C rayrs_amd/csrc)")
    dp = C.POINTER(C.c_double)
    vp = C.c_void_p
    L.rayrs_strerror.restype = C.c_char_p
    L.rayrs_strerror.argtypes = [C.c_int]
    L.rayrs_last_error.restype = C.c_char_p
    L.rayrs_objects_create.argtypes = [C.POINTER(vp)]
    L.rayrs_objects_destroy.argtypes = [vp]
    L.rayrs_objects_destroy.restype = None
    L.rayrs_objects_len.argtypes = [vp]
    L.rayrs_objects_len.restype = C.c_uint64
    mp, ep = C.POINTER(MaterialDesc), C.POINTER(EmissionDesc)
    L.rayrs_object_sphere.argtypes = [vp, C.c_double, dp, mp, ep]
    L.rayrs_object_plane.argtypes = [vp, C.c_int] + [C.c_double] * 5 + [mp, ep]
    L.rayrs_object_triangle.argtypes = [vp, dp, dp, dp, mp, ep]
    L.rayrs_object_from_triangles_f32.argtypes = [vp, vp, C.c_uint32, vp, C.c_uint32, mp, ep]
    L.rayrs_object_from_triangles_f64.argtypes = [vp, vp, C.c_uint32, vp, C.c_uint32, mp, ep]
    L.rayrs_object_from_spheres.argtypes = [vp, C.c_double, vp, C.c_uint32, mp, ep]
    L.rayrs_object_box_geom.argtypes = [vp, dp, dp, mp, ep]
    L.rayrs_scene_new.argtypes = [vp, C.c_double, C.c_double, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, vp,
                                  C.c_int, C.POINTER(vp)]
    L.rayrs_scene_destroy.argtypes = [vp]
    L.rayrs_scene_destroy.restype = None
    L.rayrs_scene_info.argtypes = [vp, C.POINTER(SceneInfo)]
    L.rayrs_scene_export_bvh.argtypes = [vp, vp, vp, vp]
    L.rayrs_scene_export_wide.argtypes = [vp, vp, vp]
    L.rayrs_scene_export_gate_tree.argtypes = [vp, vp, vp]
    L.rayrs_scene_export_hot_tree.argtypes = [vp, vp, vp]
    L.rayrs_scene_clone_to_device.argtypes = [vp, C.c_int, C.POINTER(vp)]
    L.rayrs_scene_device.argtypes = [vp]
    L.rayrs_scene_set_tuning.argtypes = [vp, C.POINTER(Tuning)]
    L.rayrs_lab_set.argtypes = [vp, C.POINTER(LabTuning)]  # private: rayrs_amd/csrc/rayrs_lab.h
    L.rayrs_lab_set.restype = C.c_int
    L.rayrs_lab_multi_rehearse.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_int, C.c_char_p, C.c_uint32]
    L.rayrs_lab_multi_rehearse.restype = C.c_int
    L.rayrs_frame_sample_chunk.argtypes = [C.c_uint32] * 4
    L.rayrs_frame_sample_chunk.restype = C.c_uint32
    L.rayrs_abi_layout.argtypes = [vp, C.c_uint32]
    L.rayrs_abi_layout.restype = C.c_uint32
    L.rayrs_abi_version.argtypes = []
    L.rayrs_abi_version.restype = C.c_uint32
    L.rayrs_render_multi.argtypes = [C.POINTER(vp), C.c_uint32, C.POINTER(CameraDesc), C.POINTER(RenderParams), vp,
                                     C.POINTER(RenderStats)]
    L.rayrs_camera_new.argtypes = [dp, dp, dp, C.c_double, C.c_double, C.c_double, C.c_uint32,
                                   C.POINTER(CameraDesc)]
    L.rayrs_render.argtypes = [vp, C.POINTER(CameraDesc), C.POINTER(RenderParams), vp, C.POINTER(RenderStats)]
    L.rayrs_render_launch.argtypes = [vp, C.POINTER(CameraDesc), C.POINTER(RenderParams), vp, vp]
    L.rayrs_render_finish.argtypes = [vp, C.POINTER(RenderStats)]
    L.rayrs_test_math.argtypes = [C.c_int, C.c_int, vp, vp, C.c_uint64, vp]
    L.rayrs_test_rng.argtypes = [C.c_int, C.c_uint64, vp, vp, vp, C.c_uint64, vp]
    L.rayrs_test_intersect.argtypes = [vp, vp, vp, C.c_uint64, C.c_int, vp, vp]
    L.rayrs_test_path_trace.argtypes = [vp, C.POINTER(CameraDesc), C.c_uint64, C.c_uint32, vp, vp, C.c_uint64, C.c_int, C.c_uint32,
                                        vp, vp, vp, vp, vp, vp]
    L.rayrs_test_material.argtypes = [C.c_int, mp, vp, vp, vp, C.c_uint64, vp, vp, vp, vp]
    L.rayrs_test_background.argtypes = [vp, vp, C.c_uint64, vp]
    L.rayrs_io_last_error.restype = C.c_char_p
    L.rayrs_buffer_free.argtypes = [vp]
    L.rayrs_buffer_free.restype = None
    u32p = C.POINTER(C.c_uint32)
    L.rayrs_ply_load.argtypes = [C.c_char_p, C.POINTER(vp), u32p, C.POINTER(vp), u32p]
    L.rayrs_ply_save.argtypes = [C.c_char_p, vp, C.c_uint32, vp, C.c_uint32, C.c_int]
    L.rayrs_obj_load.argtypes = [C.c_char_p, C.POINTER(vp), u32p, C.POINTER(vp), u32p]
    L.rayrs_obj_load_spheres.argtypes = [C.c_char_p, C.POINTER(vp), u32p]
    L.rayrs_hdr_load.argtypes = [C.c_char_p, C.POINTER(vp), u32p, u32p]
    L.rayrs_hdr_save.argtypes = [C.c_char_p, vp, C.c_uint32, C.c_uint32]
    L.rayrs_image_to_bytes.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_double, vp, C.POINTER(C.c_uint64)]
    L.rayrs_ppm_save.argtypes = [C.c_char_p, vp, C.c_uint32, C.c_uint32]
    L.rayrs_png_save.argtypes = [C.c_char_p, vp, C.c_uint32, C.c_uint32]
    for name in SYMBOLS:
        fn = getattr(L, name)
        if fn.restype is C.c_int and name not in ("rayrs_strerror",):
            fn.restype = C.c_int
    _lib = L
    return L


class RayrsError(RuntimeError):
    def __init__(self, status, where):
        L = lib()
        msg = L.rayrs_strerror(status).decode()
        extra = L.rayrs_last_error().decode() if status != -6 else L.rayrs_io_last_error().decode()
        super().__init__(f"{where}: {msg} ({status})" + (f" [{extra}]" if extra else ""))
        self.status = status


def check(status, where):
    if status != 0:
        raise RayrsError(status, where)
