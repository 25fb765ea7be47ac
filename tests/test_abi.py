"""The C-ABI library without a GPU: it loads, exports every symbol include/rayrs_hip.h
declares, validates arguments like the reference's assert!s and never aborts."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import rayrs_amd
from rayrs_amd import _ffi, procedural, scenes
from rayrs_amd.api import Axis, BvhHeuristic, Emission, Material, Object

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDRI = procedural.make_hdri(32, 16)
NR, DARK = Material.NoReflect(), Emission.Dark()


def declared_functions():
    text = open(os.path.join(ROOT, "include", "rayrs_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rayrs_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported():
    L = _ffi.lib()
    names = declared_functions()
    assert len(names) >= 20
    for name in names:
        assert hasattr(L, name), name
    assert sorted(_ffi.SYMBOLS) == names


def test_product_never_loads_the_oracle():
    """The shipped library must not depend on anything under oracle/."""
    import subprocess
    out = subprocess.run(["ldd", _ffi.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in out
    for dirpath, _, files in os.walk(os.path.join(ROOT, "rayrs_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle/" not in text and "_oracle" not in text and "librayrs_oracle" not in text, f


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_ffi, "_lib", None)
    monkeypatch.setattr(_ffi, "LIB_PATH", "/nonexistent/librayrs_hip.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _ffi.lib()


def test_strerror_and_status_codes():
    L = _ffi.lib()
    assert L.rayrs_strerror(0) == b"ok"
    assert b"assert" in L.rayrs_strerror(-1)
    assert L.rayrs_objects_create(None) == -1


@pytest.mark.parametrize("bad", [
    lambda: Object.sphere(0.0, (0, 0, 0), NR, DARK),                                   # geometry.rs:97
    lambda: Object.plane(Axis.Z, 1., -1., 1., -1., 0., NR, DARK),                      # geometry.rs:205
    lambda: Object.sphere(1.0, (0, 0, 0), Material.LambertianDiffuse((1.1, 0, 0)), DARK),  # material.rs:609
    lambda: Object.sphere(1.0, (0, 0, 0), Material.CookTorranceGlass((1, 1, 1), 0.0, 1.45), DARK),  # :865
    lambda: Object.sphere(1.0, (0, 0, 0), Material.Glass((1, 1, 1), 0.0), DARK),       # :675
    lambda: Object.sphere(1.0, (0, 0, 0), NR, Emission.new(-1.0, (1, 1, 1))),          # :1068
])
def test_object_asserts_become_value_errors(bad):
    with pytest.raises(ValueError):
        rayrs_amd.Scene([bad()], 1e-6, 1e6, BvhHeuristic.Sah(1000), HDRI, device=-1)


def test_scene_and_camera_asserts():
    s = Object.sphere(1.0, (0, 0, 0), NR, DARK)
    for zn, zf in ((-1.0, 1e6), (1.0, 1.0), (2.0, 1.0)):                               # lib.rs:234-235
        with pytest.raises(ValueError):
            rayrs_amd.Scene([s], zn, zf, BvhHeuristic.Midpoint, HDRI, device=-1)
    with pytest.raises(ValueError):                                                     # bvh.rs:229
        rayrs_amd.Scene([], 1e-6, 1e6, BvhHeuristic.Midpoint, HDRI, device=-1)
    for fov, w, h, la in ((0., 1., 1., (0, 0, 1)), (180., 1., 1., (0, 0, 1)), (90., 0., 1., (0, 0, 1)),
                          (90., 1., -1., (0, 0, 1)), (90., 1., 1., (0, 0, 0))):          # lib.rs:108-111
        with pytest.raises(ValueError):
            rayrs_amd.Camera((0, 0, 0), (0, 1, 0), la, fov, w, h, 100)


def test_camera_matches_reference_doc_test():
    cam = rayrs_amd.Camera((1, 1, 1), (0, 1, 0), (0, 0, 0), 90., 20., 10., 90)          # lib.rs:141-173
    assert cam.x_pixels() == 4580 and cam.y_pixels() == 2290
    import _oracle
    oc = _oracle.OracleCamera((1, 1, 1), (0, 1, 0), (0, 0, 0), 90., 20., 10., 90)
    for f in ("origin", "e_x", "e_y", "z"):
        assert list(getattr(cam.desc, f)) == list(getattr(oc.desc, f))


def test_host_only_scene_cannot_render_but_can_be_inspected():
    cam_args, objs, heur = scenes.diffuse_single_sphere()
    sc = rayrs_amd.Scene(objs, 1e-6, 1e6, heur, HDRI, device=-1)
    info = sc.info()
    assert info["n_objects"] == 2 and info["n_prims"] == 2 and info["n_interior"] == 0
    cam = rayrs_amd.Camera(*scenes.camera_for_resolution(cam_args, 32, 32))
    with pytest.raises(_ffi.RayrsError) as e:
        rayrs_amd.render(sc, cam, 1)
    assert e.value.status == -4


def test_mesh_index_out_of_range_is_rejected():
    verts = np.zeros((3, 3), dtype=np.float32)
    idx = np.array([[0, 1, 3]], dtype=np.uint32)
    with pytest.raises(ValueError):
        rayrs_amd.Scene(Object.from_triangles(verts, idx, NR, DARK), 1e-6, 1e6, BvhHeuristic.Midpoint, HDRI, device=-1)


def test_box_geom_keeps_the_reference_face_order():
    """lib.rs:444-505: X, XRev, ZRev, Z, YRev, Y -- and both Y faces at lower_left.y."""
    faces = Object.box_geom((0, 1, 2), (3, 4, 5), NR, DARK)
    assert [f.axis for f in faces] == [Axis.X, Axis.XRev, Axis.ZRev, Axis.Z, Axis.YRev, Axis.Y]
    assert faces[4].pos == 1.0 and faces[5].pos == 1.0


def test_ctypes_structs_have_the_c_layout():
    """rayrs_abi_layout() reports sizeof / offsetof of every public struct as the library was compiled;
    rayrs_amd/_ffi.py's ctypes mirrors must agree field for field (INTEGRATION.md's #[repr(C)] structs are
    written from the same table)."""
    L = _ffi.lib()
    n = L.rayrs_abi_layout(None, 0)
    table = (C.c_uint32 * n)()
    assert L.rayrs_abi_layout(table, n) == n
    table, pos = list(table), 1
    # the table's first word is the boundary's version: a binding that validates itself against the table alone
    # still fails when a field changes its meaning at an unchanged offset (round 5's exact_traversal -> fast_traversal)
    assert table[0] == L.rayrs_abi_version() == _ffi.ABI_VERSION
    for st in _ffi.ABI_STRUCTS:
        size, nfields = table[pos], table[pos + 1]
        offsets = table[pos + 2:pos + 2 + nfields]
        pos += 2 + nfields
        assert C.sizeof(st) == size, st.__name__
        assert len(st._fields_) == nfields, st.__name__
        assert [getattr(st, name).offset for name, _ in st._fields_] == offsets, st.__name__
    assert pos == n


def test_integration_md_quotes_the_struct_sizes():
    L = _ffi.lib()
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for st, cname in zip(_ffi.ABI_STRUCTS, ["rayrs_material", "rayrs_emission", "rayrs_camera", "rayrs_scene_info_t",
                                            "rayrs_render_params", "rayrs_render_stats", "rayrs_tuning"]):
        assert re.search(rf"{cname}\W+{C.sizeof(st)} bytes", text), f"INTEGRATION.md: {cname} is {C.sizeof(st)} bytes"


def test_frame_sample_chunk_depends_on_the_frame_only():
    """The chunk rule (rayrs_frame_sample_chunk): smallest requested * 2^k with at most 2^30 (pixel, chunk)
    items in the WHOLE frame -- the same for every rank count."""
    f = rayrs_amd.frame_sample_chunk
    assert f(2048, 2048, 1024) == 4      # configs[4]: exactly 2^30 items
    assert f(2048, 2048, 4096) == 16     # configs[3]
    assert f(1024, 1024, 512) == 4 and f(1024, 1024, 256) == 4 and f(256, 256, 64) == 4
    assert f(2049, 2048, 1024) == 8      # one more tile column tips it over
    assert f(64, 64, 4) == 0 and f(64, 64, 3) == 0   # chunk >= spp: one sequential sum, the reference's order
    assert f(64, 64, 64, requested=16) == 16
    for w, h, spp in ((2048, 2048, 1024), (4096, 4096, 4096), (333, 77, 100000)):
        c = f(w, h, spp) or spp
        assert ((w + 7) // 8) * ((h + 7) // 8) * 64 * -(-spp // c) <= 1 << 30


def test_tuning_is_validated_and_needs_no_gpu():
    cam_args, objs, heur = scenes.diffuse_single_sphere()
    sc = rayrs_amd.Scene(objs, 1e-6, 1e6, heur, HDRI, device=-1)
    sc.set_tuning(pool_slots=65536, local_pool=1)
    with pytest.raises(_ffi.RayrsError):
        sc.set_tuning(local_pool=2)
    with pytest.raises(ValueError):
        sc.set_tuning(no_such_knob=1)
    # the development knobs (rayrs_amd/csrc/rayrs_lab.h): validated too
    sc.lab_set(stack_lds=2, refill_min=40, local_reserve=64, local_segment_items=65536)
    for bad in (dict(stack_lds=65), dict(static_pct=101), dict(refill_min=65), dict(local_reserve=5),
                dict(local_segment_items=1000), dict(force_rccl=2)):
        with pytest.raises(_ffi.RayrsError):
            sc.lab_set(**bad)
    with pytest.raises(ValueError):
        sc.lab_set(no_such_knob=1)


def test_the_public_header_lists_no_experiment_selector():
    """include/rayrs_hip.h is the reference's interface for this path plus what a caller may legitimately choose; the
    kernels' development knobs live in rayrs_amd/csrc/rayrs_lab.h, which nothing outside tests/ and scripts/ needs."""
    hdr = open(os.path.join(ROOT, "include", "rayrs_hip.h")).read()
    body = hdr[hdr.index("typedef struct {\n    uint32_t pool_slots;"):hdr.index("} rayrs_tuning;")]
    assert re.findall(r"uint32_t (\w+);", body) == ["pool_slots", "local_pool"]
    for word in ("leaf_group", "trav_queries", "stream_pool", "hit_blocks_per_cu", "pipelines", "rayrs_lab", "refill_min"):
        assert not re.search(rf"\b{word}\b(?!\.h)", hdr), word
    assert not os.path.exists(os.path.join(ROOT, "rayrs_amd", "csrc", "stream_pool.hip"))
    # the device self-test hooks are not the boundary either: rayrs_amd/csrc/rayrs_selftest.h declares them
    assert "rayrs_test_" not in hdr
    priv = open(os.path.join(ROOT, "rayrs_amd", "csrc", "rayrs_selftest.h")).read() + open(os.path.join(ROOT, "rayrs_amd", "csrc", "rayrs_lab.h")).read()
    L = _ffi.lib()
    for name in _ffi.PRIVATE_SYMBOLS:
        assert re.search(rf"\b{name}\s*\(", priv), name
        assert hasattr(L, name), name


def test_abi_version_and_zero_initialised_params():
    """RAYRS_ABI_VERSION is what the library was compiled with; a zero-initialised rayrs_render_params asks for the
    reference's visit set (fast_traversal = 0), and a field out of range is refused, not interpreted."""
    L = _ffi.lib()
    hdr = open(os.path.join(ROOT, "include", "rayrs_hip.h")).read()
    assert int(re.search(r"#define RAYRS_ABI_VERSION (\d+)", hdr).group(1)) == L.rayrs_abi_version() == _ffi.ABI_VERSION >= 6
    p = _ffi.RenderParams()
    assert p.fast_traversal == 0
    assert rayrs_amd.make_params(4).fast_traversal == 0 and rayrs_amd.make_params(4, fast_traversal=True).fast_traversal == 1


def test_which_route_a_scene_takes_is_decided_at_scene_new_and_needs_no_gpu():
    """rayrs_scene_info_t.local_pool: a walk tree of at most one record (the reference's sphere scenes) is rendered by
    local_pool.hip, everything else by the streaming kernels; rayrs_tuning.local_pool = 1 switches the former off."""
    from rayrs_amd.api import Emission, Material, Object
    for fn, want in ((scenes.diffuse_single_sphere, 1), (scenes.cook_torrance_spheres_metallic, 1),
                     (scenes.material_test, 1), (lambda: scenes.mesh_scene(2), 0)):
        cam_args, objs, heur = fn()
        sc = rayrs_amd.Scene(objs, 1e-6, 1e6, heur, HDRI, device=-1)
        info = sc.info()
        assert info["local_pool"] == want, fn
        assert (info["gate_n_wide"] <= 1 and info["n_prims"] <= 16) == bool(want)
        sc.set_tuning(local_pool=1)
        assert sc.info()["local_pool"] == 0
    # seventeen primitives under one record cannot happen (4 slots x 4), but seventeen primitives can: two records
    mat = Material.LambertianDiffuse((0.5, 0.5, 0.5))
    objs = [Object.sphere(0.3, (float(i % 5), 0.3, float(i // 5)), mat, Emission.Dark()) for i in range(17)]
    sc = rayrs_amd.Scene(objs, 1e-6, 1e6, scenes.SAH_1000, HDRI, device=-1)
    assert sc.info()["n_wide"] > 1 and sc.info()["local_pool"] == 0


def test_no_environment_variable_reaches_the_product():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "rayrs_amd", "csrc")):
        for f in files:
            if f.endswith((".cpp", ".hip", ".h", ".hpp")):
                assert "getenv" not in open(os.path.join(dirpath, f)).read(), f
