"""Bvh::build (rayrs-lib/src/bvh.rs:199-389): the product's host builder
(librayrs_hip.so, host-only scene: no GPU needed) against the oracle's literal
restatement of the reference algorithm, tree for tree."""
import numpy as np
import pytest

import _oracle
import rayrs_amd
from rayrs_amd import procedural, scenes
from rayrs_amd.api import Axis, BvhHeuristic, Emission, Material, Object

HDRI = procedural.make_hdri(32, 16)
NR, DARK = Material.NoReflect(), Emission.Dark()


def same_tree(objs, heur):
    prod = rayrs_amd.Scene(objs, 1e-6, 1e6, heur, HDRI, device=-1)
    pb, pr, pp = prod.export_bvh()
    pi = prod.info()
    for builder in (0, 1):
        orc = _oracle.OracleScene(objs, 1e-6, 1e6, heur, HDRI, builder=builder)
        ob, orf, op = orc.export_bvh()
        oi = orc.flat_info()
        assert pi["n_interior"] == oi["n_interior"] and pi["n_prims"] == oi["n_prims"]
        assert pi["root_ref"] == oi["root_ref"] and pi["depth"] == oi["depth"]
        assert pi["root_box"] == oi["root_box"]
        assert np.array_equal(pr, orf) and np.array_equal(pp, op)
        assert np.array_equal(pb.view(np.uint64), ob.view(np.uint64))
        # the folded four-slot records the kernels walk
        pwb, pwr = prod.export_wide()
        owb, owr = orc.export_wide()
        assert pi["n_wide"] == oi["n_wide"] and pi["wide_root_ref"] == oi["wide_root_ref"]
        assert pi["wide_depth"] == oi["wide_depth"]
        assert np.array_equal(pwr, owr)
        used = (pwr >> 30) < 3
        assert np.array_equal(pwb[used].view(np.uint64), owb[used].view(np.uint64))
        check_fold(pb, pr, pwb, pwr, pi)
    return pi, pr, pp


def check_fold(box, ref, wbox, wref, info):
    """Independent structural check of the fold: walking the wide records reaches exactly the
    primitives of the two-child tree, every box-tested reference of that tree appears once
    behind its own box, and a direct leaf pulled up a level sits behind its parent's box."""
    if info["n_interior"] == 0:
        assert info["n_wide"] == 0 and info["wide_root_ref"] == info["root_ref"]
        return
    expect = []  # (ref, box bytes or None) reachable from binary record n, two levels at a time

    def slots_of(n):
        out = []
        for c in range(2):
            r = int(ref[n, c])
            if r >> 30 == 0:
                for g in range(2):
                    rg = int(ref[r & 0x3fffffff, g])
                    if rg >> 30 == 2:
                        out.append(((1 << 30) | (rg & 0x3fffffff), box[n, c].tobytes(), None))
                    else:
                        out.append((rg, box[r & 0x3fffffff, g].tobytes(), (rg & 0x3fffffff) if rg >> 30 == 0 else None))
            else:
                out.append((r, None if r >> 30 == 2 else box[n, c].tobytes(), None))
        return out

    seen = 0
    todo = [(info["root_ref"] & 0x3fffffff, info["wide_root_ref"] & 0x3fffffff)]
    while todo:
        n, w = todo.pop()
        seen += 1
        want = slots_of(n)
        assert [int(x) >> 30 for x in wref[w, len(want):]] == [3] * (4 - len(want))
        for k, (r, bx, sub) in enumerate(want):
            got = int(wref[w, k])
            if sub is None:
                assert got == r
            else:
                assert got >> 30 == 0
                todo.append((sub, got & 0x3fffffff))
            if bx is not None:
                assert wbox[w, k].tobytes() == bx
    assert seen == info["n_wide"]


SCENE_FNS = [scenes.diffuse_single_sphere, scenes.cook_torrance_spheres_metallic, scenes.material_test,
             lambda: scenes.mesh_scene(2), lambda: scenes.mesh_scene(3, area_light=True)]


@pytest.mark.parametrize("fn", SCENE_FNS, ids=["single_sphere", "sphere_row", "material_test", "mesh320", "mesh1280"])
@pytest.mark.parametrize("heur", [BvhHeuristic.Sah(1000), BvhHeuristic.Sah(7), BvhHeuristic.Midpoint],
                         ids=["sah1000", "sah7", "midpoint"])
def test_builder_matches_reference_algorithm(fn, heur):
    cam_args, objs, _ = fn()
    same_tree(objs, heur)


def test_random_object_soup():
    r = np.random.default_rng(4)
    objs = []
    for i in range(300):
        c = r.uniform(-5, 5, 3)
        kind = i % 3
        if kind == 0:
            objs.append(Object.sphere(float(r.uniform(0.05, 0.6)), c, NR, DARK))
        elif kind == 1:
            objs.append(Object.plane(int(r.integers(0, 6)), c[0], c[0] + 0.5, c[1], c[1] + 0.7, c[2], NR, DARK))
        else:
            objs.append(Object.triangle(c, c + r.uniform(-1, 1, 3), c + r.uniform(-1, 1, 3), NR, DARK))
    # coincident centres force the median fallback (bvh.rs:279-287)
    objs += [Object.sphere(0.3, (1.0, 1.0, 1.0), NR, DARK) for _ in range(9)]
    same_tree(objs, BvhHeuristic.Sah(1000))
    same_tree(objs, BvhHeuristic.Midpoint)


def test_midpoint_topology_documented_by_the_disabled_reference_tests():
    """bvh.rs:436-485 (commented out, stale Debug format): 8 unit spheres on a line at
    -10.5 + 3 i split into two bottom nodes of four, x-range [-11.5,-0.5] and [0.5,11.5]."""
    for axis in range(3):
        objs = []
        for i in range(8):
            c = [0.0, 0.0, 0.0]
            c[axis] = -10.5 + 3.0 * i
            objs.append(Object.sphere(1.0, c, NR, DARK))
        info, refs, prims = same_tree(objs, BvhHeuristic.Midpoint)
        assert info["n_interior"] == 1 and info["depth"] == 1
        assert list(prims) == list(range(8))
        # both children: leaf ranges of 4 behind a box test
        assert [int(x) >> 30 for x in refs[0]] == [1, 1]
        assert [(int(x) & 3) + 1 for x in refs[0]] == [4, 4]
        box, _, _ = rayrs_amd.Scene(objs, 1e-6, 1e6, BvhHeuristic.Midpoint, HDRI, device=-1).export_bvh()
        assert box[0, 0, 2 * axis] == -11.5 and box[0, 0, 2 * axis + 1] == -0.5
        assert box[0, 1, 2 * axis] == 0.5 and box[0, 1, 2 * axis + 1] == 11.5


def test_small_scenes_are_one_bottom_node():
    for n in (1, 2, 4):
        objs = [Object.sphere(1.0, (3.0 * i, 0, 0), NR, DARK) for i in range(n)]
        info, refs, prims = same_tree(objs, BvhHeuristic.Sah(1000))
        assert info["n_interior"] == 0 and info["root_ref"] >> 30 == 1 and (info["root_ref"] & 3) + 1 == n


def test_five_objects_split_with_single_leaf_children():
    """len > 4 splits; a LEFT side with one object becomes a direct leaf with no box test
    (bvh.rs:294-303).  A right side of one object (ind == len-1) falls back to the median
    instead (bvh.rs:279-287), a quirk the builder keeps."""
    objs = [Object.sphere(0.5, (float(x), 0, 0), NR, DARK) for x in (0, 50, 51, 52, 53)]
    info, refs, prims = same_tree(objs, BvhHeuristic.Sah(1000))
    assert info["n_interior"] == 1 and [int(x) >> 30 for x in refs[0]] == [2, 1]
    objs = [Object.sphere(0.5, (float(x), 0, 0), NR, DARK) for x in (0, 1, 2, 3, 50)]
    info, refs, prims = same_tree(objs, BvhHeuristic.Sah(1000))
    assert [int(x) >> 30 for x in refs[0]] == [1, 1] and [(int(x) & 3) + 1 for x in refs[0]] == [2, 3]


def test_compact_layout_only_when_exact_in_f32():
    cam, objs, heur = scenes.mesh_scene(2)
    assert rayrs_amd.Scene(objs, 1e-6, 1e6, heur, HDRI, device=-1).info()["compact"] == 1
    cam, objs, heur = scenes.cook_torrance_spheres_metallic()  # 2.2 * k is not an f32
    assert rayrs_amd.Scene(objs, 1e-6, 1e6, heur, HDRI, device=-1).info()["compact"] == 0
    verts, idx = procedural.blob_mesh(1)
    v64 = verts.astype(np.float64) + 1e-9  # no longer f32 values
    objs = Object.from_triangles(v64, idx, NR, DARK)
    assert rayrs_amd.Scene(objs, 1e-6, 1e6, heur, HDRI, device=-1).info()["compact"] == 0


def test_large_mesh_builder_agreement():
    cam, objs, heur = scenes.mesh_scene(5)  # 20480 triangles
    prod = rayrs_amd.Scene(objs, 1e-6, 1e6, heur, HDRI, device=-1)
    orc = _oracle.OracleScene(objs, 1e-6, 1e6, heur, HDRI, builder=1)
    pb, pr, pp = prod.export_bvh()
    ob, orf, op = orc.export_bvh()
    assert np.array_equal(pr, orf) and np.array_equal(pp, op) and np.array_equal(pb, ob)
    pwb, pwr = prod.export_wide()
    owb, owr = orc.export_wide()
    assert np.array_equal(pwr, owr) and np.array_equal(pwb[(pwr >> 30) < 3], owb[(owr >> 30) < 3])
    check_fold(pb, pr, pwb, pwr, prod.info())


def test_largest_boxes_lead_the_wide_records():
    """scene_host.cpp front_largest: the 256 wide records with the largest boxes come first,
    largest first, the others keep their depth-first order (the traversal kernel holds the
    first of them in LDS)."""
    cam, objs, heur = scenes.mesh_scene(5)
    prod = rayrs_amd.Scene(objs, 1e-6, 1e6, heur, HDRI, device=-1)
    box, ref = prod.export_wide()
    tested = (ref >> 30) < 2
    lo = np.where(tested[:, :, None], box[:, :, 0::2], np.inf).min(axis=1)
    hi = np.where(tested[:, :, None], box[:, :, 1::2], -np.inf).max(axis=1)
    e = hi - lo
    area = 2.0 * (e[:, 0] * e[:, 1] + e[:, 1] * e[:, 2] + e[:, 0] * e[:, 2])
    assert len(area) > 1000
    front, rest = area[:256], area[256:]
    assert np.all(front[:-1] >= front[1:])
    assert front[-1] >= rest.max()
    # behind the front, a record's children still follow it (depth-first order survives the move)
    kids = ref[256:][(ref[256:] >> 30) == 0] & 0x3fffffff
    owner = np.repeat(np.arange(256, len(ref)), 4).reshape(-1, 4)[(ref[256:] >> 30) == 0]
    later = kids >= 256
    assert np.all(kids[later] > owner[later])
