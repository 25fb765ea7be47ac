"""Bvh::build (rayrs-lib/src/bvh.rs:199-389): the product's host builder
(librayrs_hip.so, host-only scene: no GPU needed) against the oracle's literal
restatement of the reference algorithm, tree for tree -- and the tree the kernels
actually walk, which is the product's own (built for traversal speed over the
reference's leaf groups), against the properties that make it return the
reference's hits."""
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
    # the four-slot records the kernels walk are the product's own trees over the reference's groups / primitives
    check_walk_trees(prod, orc, pi["n_prims"])
    return pi, pr, pp


def reference_groups(box, ref, info):
    """The groups of the reference's tree, from the two-child export alone: every run of 1..4 leaves
    that share a parent Node, with the box that gates them -- the Node's own box, which for a
    direct leaf (bvh.rs:297, :302: no box of its own) is the box of the record it hangs under.
    Returns {range reference: box bytes} in depth-first order."""
    groups = {}
    root_box = np.array(info["root_box"], dtype=np.float64)
    if info["n_interior"] == 0:
        return {int(info["root_ref"]): root_box.tobytes()}
    todo = [(info["root_ref"] & 0x3fffffff, root_box)]
    while todo:
        n, gate = todo.pop()
        for c in (1, 0):
            r = int(ref[n, c])
            kind = r >> 30
            if kind == 0:
                todo.append((r & 0x3fffffff, box[n, c]))
            else:
                g = (1 << 30) | (r & 0x3fffffff)
                assert g not in groups
                groups[g] = (gate if kind == 2 else box[n, c]).tobytes()
    return groups


LEAF_MARGIN = 2.0 ** -6  # scene_host.cpp


def tight_box(prim_box, gate):
    """scene_host.cpp tight_box, restated: the primitive's box widened by LEAF_MARGIN of its largest extent,
    rounded outwards to f32, clipped to the gating box."""
    ext = max(prim_box[1] - prim_box[0], prim_box[3] - prim_box[2], prim_box[5] - prim_box[4])
    m = ext * LEAF_MARGIN
    out = np.zeros(6)
    for a in range(3):
        lo, hi = prim_box[2 * a] - m, prim_box[2 * a + 1] + m
        with np.errstate(over="ignore"):
            lf, hf = np.float32(lo), np.float32(hi)
        if float(lf) > lo:
            lf = np.nextafter(lf, np.float32(-np.inf))
        if float(hf) < hi:
            hf = np.nextafter(hf, np.float32(np.inf))
        out[2 * a] = float(lf) if float(lf) > gate[2 * a] else gate[2 * a]
        out[2 * a + 1] = float(hf) if float(hf) < gate[2 * a + 1] else gate[2 * a + 1]
    return out


def check_walk_tree(box, ref, wbox, wref, info, prim_boxes=None, gate=False, hot=False):
    """What makes a walk tree legal (scene_host.cpp build_walk_trees), checked from the outside: every interior
    slot's box is the union of the boxes below it (so a ray that misses it misses every leaf box inside); unused
    slots are marked; every record is reached once; the stack bound holds; and
      gate=True (the tree the default walk reads): every group of the reference's tree sits in exactly one leaf slot
        behind exactly its gating box -- the tree reaches what the reference reaches;
      gate=False (the default tree): every primitive sits alone in exactly one leaf slot, behind its own bounding
        box (prim_boxes[p]: Object::bbox of the object behind primitive p) widened by LEAF_MARGIN and clipped to its
        group's gating box -- inside the gating box, so it reaches nothing the reference does not, and around the
        primitive with the margin to spare;
      hot=True (with gate=True: rayrs_scene_export_hot_tree, what the default walk reads on a scene with a hot group):
        as gate=True, but ONE group is not in the records -- the hot group of rayrs_scene_info_t.hot_*: it must be one
        of the reference's groups, behind exactly its gating box, the one with the largest box, covering at least a
        quarter of the root Node's box (scene_host.cpp pick_hot_group): the tree's groups and that one are the groups."""
    groups = reference_groups(box, ref, info)
    pre = "hot_" if hot else ("gate_" if gate else "wide_")
    n_wide = info["hot_n_wide" if hot else ("gate_n_wide" if gate else "n_wide")]
    hot_ref = None
    if hot:
        assert gate and info["hot_count"] >= 1
        hot_ref = (1 << 30) | (info["hot_first"] << 2) | (info["hot_count"] - 1)
        assert hot_ref in groups and np.array(info["hot_box"]).tobytes() == groups[hot_ref]

        def area(bb):
            b = np.frombuffer(bb, dtype=np.float64)
            x, y, z = b[1] - b[0], b[3] - b[2], b[5] - b[4]
            return 2.0 * (x * y + y * z + x * z)
        areas = {g: area(bb) for g, bb in groups.items()}
        assert areas[hot_ref] == max(areas.values()) and areas[hot_ref] >= 0.25 * area(np.array(info["root_box"]).tobytes())
        assert len(groups) >= 8
    if info["n_interior"] == 0:  # one bottom Node: both trees are the root group behind the root box
        assert n_wide == 0 and info[pre + "root_ref"] == info["root_ref"] and info[pre + "depth"] == 0
        return
    assert info[pre + "root_ref"] >> 30 == 0
    gate_of = {}  # primitive -> its group's gating box
    for g, bb in groups.items():
        first, count = (g & 0x3fffffff) >> 2, (g & 3) + 1
        for p in range(first, first + count):
            gate_of[p] = np.frombuffer(bb, dtype=np.float64)
    seen_leaves = set()
    seen_records = set()

    def visit(w):  # -> (union box of the record's slots, stack entries needed below)
        assert w not in seen_records
        seen_records.add(w)
        lo, hi = np.full(3, np.inf), np.full(3, -np.inf)
        used, below = 0, 0
        for k in range(4):
            r = int(wref[w, k])
            kind = r >> 30
            if kind == 3:
                assert all(int(x) >> 30 == 3 for x in wref[w, k:])  # unused slots trail
                break
            used += 1
            b = wbox[w, k]
            if kind == 1 and gate:
                assert r in groups and r not in seen_leaves
                seen_leaves.add(r)
                assert b.tobytes() == groups[r]
            elif kind == 1:
                assert r & 3 == 0  # one primitive
                p = (r & 0x3fffffff) >> 2
                assert p in gate_of and p not in seen_leaves
                seen_leaves.add(p)
                g, pb = gate_of[p], prim_boxes[p]
                assert b.tobytes() == tight_box(pb, g).tobytes()
                # what the construction is for (boxes whose bounds are not numbers aside)
                if np.isfinite(pb).all() and np.isfinite(g).all():
                    assert (b[0::2] >= g[0::2]).all() and (b[1::2] <= g[1::2]).all()
                    assert (b[0::2] <= np.maximum(pb[0::2], g[0::2])).all() and (b[1::2] >= np.minimum(pb[1::2], g[1::2])).all()
            else:
                assert kind == 0
                (clo, chi), need = visit(r & 0x3fffffff)
                assert np.array_equal(b[0::2], clo) and np.array_equal(b[1::2], chi)
                below = max(below, need)
            lo, hi = np.minimum(lo, b[0::2]), np.maximum(hi, b[1::2])
        assert used >= 2
        return (lo, hi), used - 1 + below

    import sys
    sys.setrecursionlimit(max(sys.getrecursionlimit(), 10000))
    (lo, hi), need = visit(info[pre + "root_ref"] & 0x3fffffff)
    if hot:
        assert hot_ref not in seen_leaves
        seen_leaves.add(hot_ref)
        hb = np.array(info["hot_box"])
        lo, hi = np.minimum(lo, hb[0::2]), np.maximum(hi, hb[1::2])
    assert seen_leaves == (set(groups) if gate else set(gate_of)) and len(seen_records) == n_wide
    assert need == info[pre + "depth"]
    root = np.array(info["root_box"])
    if gate:
        assert np.array_equal(lo, root[0::2]) and np.array_equal(hi, root[1::2])
    else:
        assert (lo >= root[0::2]).all() and (hi <= root[1::2]).all()


def check_walk_trees(prod, orc, n_objects):
    pb, pr, pp = prod.export_bvh()
    info = prod.info()
    check_walk_tree(pb, pr, *prod.export_gate_tree(), info, gate=True)
    check_walk_tree(pb, pr, *prod.export_wide(), info, prim_boxes=orc.object_boxes(n_objects)[pp])
    if info["hot_count"]:
        check_walk_tree(pb, pr, *prod.export_hot_tree(), info, gate=True, hot=True)
    else:
        with pytest.raises(rayrs_amd._ffi.RayrsError):
            prod.export_hot_tree()
    return info["hot_count"]


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
    check_walk_trees(prod, orc, prod.info()["n_prims"])


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


def degenerate_objects():
    """Primitives whose own boxes are points, flat, huge or beyond f32, beside a floor and forty slivers."""
    r = np.random.default_rng(12)
    objs = [Object.plane(Axis.Y, -25.0, 25.0, -25.0, 25.0, 0.0, NR, DARK)]
    p = np.array([0.5, 1.0, -0.25])
    objs.append(Object.triangle(p, p, p, NR, DARK))                                  # a point
    objs.append(Object.triangle(p, p + (1.0, 0.0, 0.0), p + (2.0, 0.0, 0.0), NR, DARK))  # a segment
    objs.append(Object.triangle((0.0, 2.0, 0.0), (1.0, 2.0, 0.0), (0.0, 2.0, 1.0), NR, DARK))  # flat in y
    objs.append(Object.sphere(1e-300, (-1.0, 1.0, 1.0), NR, DARK))
    objs.append(Object.sphere(1e30, (0.0, 2e30, 0.0), NR, DARK))                     # far beyond everything else
    objs.append(Object.triangle((1e39, 0.0, 0.0), (1e39, 1e39, 0.0), (1e39, 0.0, 1e39), NR, DARK))  # beyond f32's range
    for i in range(40):
        c = r.uniform(-3, 3, 3)
        objs.append(Object.triangle(c, c + r.uniform(-1, 1, 3) * 1e-9, c + r.uniform(-1, 1, 3), NR, DARK))  # slivers
    return objs


def test_leaf_boxes_of_degenerate_and_extreme_primitives():
    """The default tree's leaf boxes (scene_host.cpp tight_box) for primitives whose own boxes are points, flat, huge or
    beyond f32: every one must still lie inside its gating box and around the primitive's box, bit for bit as restated
    here -- and the oracle's walk over both trees must return the recursion's hits."""
    r = np.random.default_rng(13)
    objs = degenerate_objects()
    for heur in (BvhHeuristic.Sah(1000), BvhHeuristic.Midpoint):
        prod = rayrs_amd.Scene(objs, 1e-6, 1e6, heur, HDRI, device=-1)
        orc = _oracle.OracleScene(objs, 1e-6, 1e6, heur, HDRI)
        check_walk_trees(prod, orc, prod.info()["n_prims"])
        o = r.uniform(-4, 4, (4000, 3))
        d = r.normal(size=(4000, 3))
        d[:500] = (0.5, 1.0, -0.25) - o[:500]   # at the point and the segment
        rt, robj = orc.intersect_many(o, d, 1e-6, 1e6, traversal=0)
        wt, wobj = orc.use_walk_tree(prod).intersect_many(o, d, 1e-6, 1e6, traversal=2)
        assert np.array_equal(wobj, robj) and np.array_equal(wt.view(np.uint64), rt.view(np.uint64))
        try:
            _oracle.set_cull_margin(float("inf"))
            xt, xobj = orc.use_walk_tree(prod, gate=True).intersect_many(o, d, 1e-6, 1e6, traversal=2)
        finally:
            _oracle.set_cull_margin(2.0 ** -10)
        assert np.array_equal(xobj, robj) and np.array_equal(xt.view(np.uint64), rt.view(np.uint64))
        assert (robj >= 0).sum() > 1000
