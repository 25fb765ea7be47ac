"""Mesh and image files either side of the hot path, through the C ABI
(rayrs_amd/csrc/host_io.cpp): PLY / OBJ ingest, Radiance .hdr in and out, the
reference's 8-bit output conversion (image.rs:193-222) and PPM / PNG writers."""
import ctypes as C

import numpy as np

from . import _ffi


def _take(ptr, count, dtype):
    L = _ffi.lib()
    n = int(count)
    arr = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(n * np.dtype(dtype).itemsize,)) if n else None
    out = np.frombuffer(arr.tobytes(), dtype=dtype).copy() if n else np.zeros(0, dtype=dtype)
    L.rayrs_buffer_free(ptr)
    return out


def load_ply(path):
    """-> (verts (n,3) float32, idx (m,3) uint32); polygons are fan-triangulated."""
    L = _ffi.lib()
    v, i = C.c_void_p(), C.c_void_p()
    nv, nt = C.c_uint32(), C.c_uint32()
    _ffi.check(L.rayrs_ply_load(str(path).encode(), C.byref(v), C.byref(nv), C.byref(i), C.byref(nt)), "rayrs_ply_load")
    return _take(v, nv.value * 3, np.float32).reshape(-1, 3), _take(i, nt.value * 3, np.uint32).reshape(-1, 3)


def save_ply(path, verts, idx, binary=True):
    verts = np.ascontiguousarray(verts, dtype=np.float32)
    idx = np.ascontiguousarray(idx, dtype=np.uint32)
    _ffi.check(_ffi.lib().rayrs_ply_save(str(path).encode(), verts.ctypes.data, verts.shape[0], idx.ctypes.data,
                                         idx.shape[0], 1 if binary else 0), "rayrs_ply_save")


def load_obj(path):
    """wavefront_obj::load_obj_file -> (verts (n,3) float64, idx (m,3) uint32)."""
    L = _ffi.lib()
    v, i = C.c_void_p(), C.c_void_p()
    nv, nt = C.c_uint32(), C.c_uint32()
    _ffi.check(L.rayrs_obj_load(str(path).encode(), C.byref(v), C.byref(nv), C.byref(i), C.byref(nt)), "rayrs_obj_load")
    return _take(v, nv.value * 3, np.float64).reshape(-1, 3), _take(i, nt.value * 3, np.uint32).reshape(-1, 3)


def load_obj_spheres(path, radius, mat=None, emission=None):
    """wavefront_obj::load_obj_file_spheres(filename, radius) (wavefront_obj.rs:46-64): one sphere per `v` line.
    -> centres (n,3) float64; with a material and an emission, the Vec<Object> Object::from_spheres makes of them."""
    L = _ffi.lib()
    c, n = C.c_void_p(), C.c_uint32()
    _ffi.check(L.rayrs_obj_load_spheres(str(path).encode(), C.byref(c), C.byref(n)), "rayrs_obj_load_spheres")
    centers = _take(c, n.value * 3, np.float64).reshape(-1, 3)
    if mat is None:
        return centers
    from .api import Object
    return Object.from_spheres(float(radius), centers, mat, emission)


def load_hdr(path):
    """-> (H, W, 3) float32, what HdrDecoder::read_image_hdr yields (main.rs:36-41)."""
    L = _ffi.lib()
    p = C.c_void_p()
    w, h = C.c_uint32(), C.c_uint32()
    _ffi.check(L.rayrs_hdr_load(str(path).encode(), C.byref(p), C.byref(w), C.byref(h)), "rayrs_hdr_load")
    return _take(p, w.value * h.value * 3, np.float32).reshape(h.value, w.value, 3)


def save_hdr(path, rgb):
    rgb = np.ascontiguousarray(rgb, dtype=np.float32)
    _ffi.check(_ffi.lib().rayrs_hdr_save(str(path).encode(), rgb.ctypes.data, rgb.shape[1], rgb.shape[0]),
               "rayrs_hdr_save")


def to_raw_bytes(rgb, gamma=1.0 / 2.2):
    """Image::to_raw_bytes (image.rs:193-222) -> ((H, W, 3) uint8, {'clamped','nan','negative'})."""
    rgb = np.ascontiguousarray(rgb, dtype=np.float32)
    out = np.zeros(rgb.shape, dtype=np.uint8)
    counts = (C.c_uint64 * 3)()
    _ffi.check(_ffi.lib().rayrs_image_to_bytes(rgb.ctypes.data, rgb.shape[1], rgb.shape[0], float(gamma),
                                               out.ctypes.data, counts), "rayrs_image_to_bytes")
    return out, {"clamped": counts[0], "nan": counts[1], "negative": counts[2]}


def save_ppm(path, bytes_rgb):
    b = np.ascontiguousarray(bytes_rgb, dtype=np.uint8)
    _ffi.check(_ffi.lib().rayrs_ppm_save(str(path).encode(), b.ctypes.data, b.shape[1], b.shape[0]), "rayrs_ppm_save")


def save_png(path, bytes_rgb):
    b = np.ascontiguousarray(bytes_rgb, dtype=np.uint8)
    _ffi.check(_ffi.lib().rayrs_png_save(str(path).encode(), b.ctypes.data, b.shape[1], b.shape[0]), "rayrs_png_save")
