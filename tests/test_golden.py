"""Committed golden vectors (tests/golden/golden_v1.npz, made by make_golden.py).

CPU: the oracle must still reproduce every vector bit for bit (any change of the
oracle or of the numeric contract shows up here).  GPU: the kernels, called
through the C ABI, must reproduce the same vectors bit for bit."""
import ctypes as C
import os

import numpy as np
import pytest

import _oracle
from golden import make_golden as G
from rayrs_amd import scenes

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_v1.npz"))
HDRI = GOLD["hdri"]


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def test_fixture_inputs_are_reproducible():
    from rayrs_amd import procedural
    assert np.array_equal(HDRI, procedural.make_hdri(*G.HDRI_SHAPE))


@pytest.mark.parametrize("name", list(G.MATERIALS))
def test_oracle_material_vectors(name):
    n, v, k = G.material_inputs(192, 7)
    sc, col, dr, nd = _oracle.material_evaluate(G.MATERIALS[name], n, v, k)
    assert np.array_equal(sc, GOLD[f"mat/{name}/scattered"])
    assert np.array_equal(nd, GOLD[f"mat/{name}/draws"])
    assert np.array_equal(bits(col), bits(GOLD[f"mat/{name}/color"]))
    assert np.array_equal(bits(dr), bits(GOLD[f"mat/{name}/dir"]))


@pytest.mark.parametrize("name", list(G.SCENES))
def test_oracle_intersections_and_frame(name):
    cam_args, objs, heur = G.SCENES[name]()
    o, d = G.rays(256, 11)
    import rayrs_amd
    prod = rayrs_amd.Scene(objs, 1e-6, 1e6, heur, HDRI, device=-1)  # host only: the tree the kernels walk
    for builder in (0, 1):  # literal reference builder and the swept one
        osc = _oracle.OracleScene(objs, 1e-6, 1e6, heur, HDRI, builder=builder).use_walk_tree(prod)
        for trav in (0, 1, 2):  # recursive reference traversal, the ordered one, the kernel's walk
            t, obj = osc.intersect_many(o, d, 1e-6, 1e6, traversal=trav)
            assert np.array_equal(obj, GOLD[f"isect/{name}/obj"])
            assert np.array_equal(bits(t), bits(GOLD[f"isect/{name}/t"]))
    ocam = _oracle.OracleCamera(*scenes.camera_for_resolution(cam_args, G.FRAME["w"], G.FRAME["h"]))
    img, st = osc.render(ocam, G.FRAME["spp"], G.FRAME["max_bounces"], seed=G.FRAME["seed"], traversal=1)
    assert st["rays"] == int(GOLD[f"frame/{name}/rays"][0])
    assert np.array_equal(bits(img), bits(GOLD[f"frame/{name}/rgb"]))
    img2, st2 = osc.render(ocam, G.FRAME["spp"], G.FRAME["max_bounces"], seed=G.FRAME["seed"], traversal=2)
    assert st2["rays"] == st["rays"] and np.array_equal(bits(img2), bits(img))


@pytest.mark.parametrize("name", list(G.SCENES))
def test_oracle_path_traces(name):
    """64 seeded samples of the golden frame, bounce by bounce: object, t, throughput, draw index."""
    cam_args, objs, heur = G.SCENES[name]()
    osc = _oracle.OracleScene(objs, 1e-6, 1e6, heur, HDRI, builder=1)
    ocam = _oracle.OracleCamera(*scenes.camera_for_resolution(cam_args, G.FRAME["w"], G.FRAME["h"]))
    pix, sam = G.trace_samples(name)
    tr = osc.path_traces(ocam, pix, sam, G.FRAME["seed"], G.FRAME["max_bounces"], G.TRACE["cap"], traversal=0)
    assert int(tr["n"].max()) <= G.TRACE["cap"] and int(tr["n"].min()) >= 1 and int(tr["n"].max()) >= 3
    for k, v in tr.items():
        g = GOLD[f"trace/{name}/{k}"]
        assert np.array_equal(bits(v), bits(g)) if v.dtype == np.float64 else np.array_equal(v, g), k


def test_oracle_background_vectors():
    cam_args, objs, heur = scenes.diffuse_single_sphere()
    osc = _oracle.OracleScene(objs, 1e-6, 1e6, heur, HDRI)
    assert np.array_equal(bits(osc.background(GOLD["background/dirs"])), bits(GOLD["background/rgb"]))
    # integral texel coordinates give black (all four weights zero), SURVEY 7(h)
    assert np.all(GOLD["background/rgb"][2] == 0.0) or np.all(GOLD["background/rgb"][3] == 0.0)


# ------------------------------------------------------------------ GPU

@pytest.mark.gpu
@pytest.mark.parametrize("name", list(G.MATERIALS))
def test_gpu_material_vectors(name):
    from rayrs_amd import _ffi
    n, v, k = G.material_inputs(192, 7)
    cnt = len(k)
    sc = np.zeros(cnt, dtype=np.int32)
    col = np.zeros((cnt, 3))
    dr = np.zeros((cnt, 3))
    nd = np.zeros(cnt, dtype=np.uint32)
    m = G.MATERIALS[name].desc()
    _ffi.check(_ffi.lib().rayrs_test_material(0, C.byref(m), n.ctypes.data, v.ctypes.data, k.ctypes.data, cnt,
                                              sc.ctypes.data, col.ctypes.data, dr.ctypes.data, nd.ctypes.data),
               "rayrs_test_material")
    hit = GOLD[f"mat/{name}/scattered"] == 1
    assert np.array_equal(sc, GOLD[f"mat/{name}/scattered"])
    assert np.array_equal(nd, GOLD[f"mat/{name}/draws"])
    assert np.array_equal(bits(col[hit]), bits(GOLD[f"mat/{name}/color"][hit]))
    assert np.array_equal(bits(dr[hit]), bits(GOLD[f"mat/{name}/dir"][hit]))


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(G.SCENES))
def test_gpu_path_traces(name):
    """The golden traces reproduced on the GPU, lane by lane, by both walks (rayrs_selftest.h rayrs_test_path_trace:
    the device functions of the path kernels run as one loop per lane): every bounce's object, t bits, throughput
    bits and draw index, and the sample's radiance."""
    import rayrs_amd
    from rayrs_amd import _ffi
    cam_args, objs, heur = G.SCENES[name]()
    scene = rayrs_amd.Scene(objs, 1e-6, 1e6, heur, HDRI, device=0)
    cam = rayrs_amd.Camera(*scenes.camera_for_resolution(cam_args, G.FRAME["w"], G.FRAME["h"]))
    pix, sam = G.trace_samples(name)
    k, cap = len(pix), G.TRACE["cap"]
    packed = np.ascontiguousarray((pix[:, 0] << 16) | pix[:, 1], dtype=np.uint32)
    sam = np.ascontiguousarray(sam, dtype=np.uint32)
    for exact in (1, 0):
        n = np.zeros(k, dtype=np.uint32); obj = np.zeros((k, cap), dtype=np.int64); t = np.zeros((k, cap))
        thr = np.zeros((k, cap, 3)); draw = np.zeros((k, cap), dtype=np.uint32); rgb = np.zeros((k, 3))
        _ffi.check(scene._L.rayrs_test_path_trace(scene._h, C.byref(cam.desc), G.FRAME["seed"], G.FRAME["max_bounces"],
                                                  packed.ctypes.data, sam.ctypes.data, k, exact, cap, n.ctypes.data,
                                                  obj.ctypes.data, t.ctypes.data, thr.ctypes.data, draw.ctypes.data,
                                                  rgb.ctypes.data), "rayrs_test_path_trace")
        for key, v in (("n", n), ("obj", obj), ("draw", draw)):
            assert np.array_equal(v, GOLD[f"trace/{name}/{key}"]), (exact, key)
        for key, v in (("t", t), ("thr", thr), ("rgb", rgb)):
            assert np.array_equal(bits(v), bits(GOLD[f"trace/{name}/{key}"])), (exact, key)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(G.SCENES))
def test_gpu_intersections_and_frame(name):
    import rayrs_amd
    from rayrs_amd import _ffi
    cam_args, objs, heur = G.SCENES[name]()
    scene = rayrs_amd.Scene(objs, 1e-6, 1e6, heur, HDRI, device=0)
    o, d = G.rays(256, 11)
    t = np.zeros(len(o))
    obj = np.zeros(len(o), dtype=np.int64)
    _ffi.check(scene._L.rayrs_test_intersect(scene._h, o.ctypes.data, d.ctypes.data, len(o), 0, t.ctypes.data,
                                             obj.ctypes.data), "rayrs_test_intersect")
    assert np.array_equal(obj, GOLD[f"isect/{name}/obj"])
    assert np.array_equal(bits(t), bits(GOLD[f"isect/{name}/t"]))
    cam = rayrs_amd.Camera(*scenes.camera_for_resolution(cam_args, G.FRAME["w"], G.FRAME["h"]))
    img, st = rayrs_amd.render(scene, cam, G.FRAME["spp"], G.FRAME["max_bounces"], seed=G.FRAME["seed"], out_f64=True)
    assert st["rays"] == int(GOLD[f"frame/{name}/rays"][0])
    assert np.array_equal(bits(img), bits(GOLD[f"frame/{name}/rgb"]))


@pytest.mark.gpu
def test_gpu_background_vectors():
    import rayrs_amd
    from rayrs_amd import _ffi
    cam_args, objs, heur = scenes.diffuse_single_sphere()
    scene = rayrs_amd.Scene(objs, 1e-6, 1e6, heur, HDRI, device=0)
    d = np.ascontiguousarray(GOLD["background/dirs"])
    out = np.zeros_like(d)
    _ffi.check(scene._L.rayrs_test_background(scene._h, d.ctypes.data, len(d), out.ctypes.data),
               "rayrs_test_background")
    assert np.array_equal(bits(out), bits(GOLD["background/rgb"]))
