"""CPU-only fuzz of the claims the traversal kernel rests on.  The reference's recursion (oracle traversal=0:
BvhTree::intersect, bvh.rs:391-415) tests every primitive whose gating box the ray enters and never culls.  Against it,
on the product's own trees (host-only scene, device = -1), walked by the oracle with the kernel's rules (traversal=2):
  exact     the gate tree with no culling (the product's DEFAULT walk): the reference's visit set by
            construction -- REQUIRED to match on every ray of every family;
  default   (the column keeps its round-4 name) the FAST walk, rayrs_render_params.fast_traversal: the tree of single
            primitives behind their widened boxes, slots entered beyond closest_t * (1 + 2^-10) culled: two bets on the
            reference's arithmetic (include/rayrs_hip.h).  REQUIRED to match on the general family and on grazing rays
            1e-7 rad and more off the plane from origins within 8 root-box diagonals AND 4096 small-primitive sizes of
            the scene -- beyond that a frame's camera gets the default walk whatever was asked (abi.cpp camera_is_far),
            and bounce rays start on the scene; counted and reported elsewhere;
  leaves    the fast walk's tree with no culling: which of its mismatches are the leaf boxes' alone.
The geometry is where Moeller-Trumbore is least accurate: sliver triangles, nearly coplanar tessellated sheets whose
group boxes are almost flat (hits sit on box faces), scene scales 1e-3 .. 1e3, origins up to 1e6 scene sizes away, and
two families of directions:
  general   elevations 1e-7 .. 1 rad over the sheet's mean plane, axis-aligned directions (0 * inf in the slab test)
  grazing   IN the plane of a chosen triangle plus 1e-13 .. 1e-3 of its normal, aimed at a point inside it
            (near: the origin within 8 root-box diagonals and 4096 small-primitive sizes of the root box -- where the fast
            walk makes its bets; far: beyond, where the product takes the default walk)
Culling loses a hit when a primitive's computed t lies more than the margin in front of a box around it (the error of
t grows like eps * distance / triangle size / angle: at 1e-9 rad and 5000 triangle sizes it reaches 2^-10).  A leaf box
loses one when the reference's own test accepts a hit on a primitive the ray passes beside by more than 1/64 of its
size (the hit POINT moves by eps * distance / angle: from 1e4 scene sizes away at 1e-7 rad).
Also measures the cull margin itself: the largest (box entry - t) / t over all accepted hits and the boxes on their
root paths (oracle: orc_cull_margin_probe), per family.

usage: python scripts/fuzz_traversal.py [rays_in_millions=10] [first_seed=1]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import _oracle, rayrs_amd
from rayrs_amd.api import BvhHeuristic, Emission, Material, Object

MAT = Material.LambertianDiffuse((0.5, 0.5, 0.5))


def sheet(rng, n, tilt, sliver):
    """(n x n) quads of a nearly planar sheet y = tilt * (ax + bz) + c as f32 vertices; `sliver` squeezes every
    other column of vertices towards its neighbour (aspect ratios up to 1 / sliver)."""
    u = np.linspace(-1.0, 1.0, n + 1)
    x, z = np.meshgrid(u, u, indexing="ij")
    x = x.copy()
    x[1::2, :] = x[0:-1:2, :][: x[1::2, :].shape[0]] + sliver * (2.0 / n)
    a, b = rng.normal(size=2)
    y = tilt * (a * x + b * z) + rng.uniform(-0.5, 0.5)
    verts = np.stack([x, y, z], axis=-1).reshape(-1, 3).astype(np.float32)
    idx = []
    for i in range(n):
        for j in range(n):
            p = i * (n + 1) + j
            idx += [(p, p + 1, p + n + 1), (p + 1, p + n + 2, p + n + 1)]
    return verts, np.array(idx, dtype=np.uint32)


def scene_for(seed):
    rng = np.random.default_rng(seed)
    kind = seed % 4
    tilt = [0.0, 1e-6, 1e-3, 0.3][kind] if seed % 8 < 4 else float(10.0 ** rng.uniform(-7, -1))
    sliver = float(10.0 ** rng.uniform(-6, 0)) if seed % 3 else 1.0
    n = int(rng.integers(6, 40))
    if seed % 11 == 0:
        n = int(rng.integers(250, 400))  # fine tessellations: a camera 8 diagonals out stands thousands of primitive sizes away
    verts, idx = sheet(rng, n, tilt, min(sliver, 1.0))
    scale = float(10.0 ** rng.uniform(-3, 3)) if seed % 5 == 0 else 1.0
    verts = (verts * scale).astype(np.float32)
    objs = Object.from_triangles(verts, idx, MAT, Emission.Dark())
    if seed % 7 == 0:  # a second sheet crossing the first: coincident / abutting hits
        v2, i2 = sheet(rng, max(4, min(n, 60) // 2), tilt * 3.0 + 1e-4, 1.0)
        objs += Object.from_triangles((v2 * scale).astype(np.float32), i2, MAT, Emission.Dark())
    if seed % 13 == 0:  # mixed scales: a floor fifty sheets wide under it (two triangles), as the benchmark scenes have
        fl = np.array([[-25, -0.5, -25], [25, -0.5, -25], [25, -0.5, 25], [-25, -0.5, 25]], dtype=np.float64) * scale
        objs += Object.from_triangles(fl.astype(np.float32), np.array([[0, 1, 2], [0, 2, 3]], dtype=np.uint32), MAT, Emission.Dark())
    heur = BvhHeuristic.Sah(int(rng.choice([4, 32, 1000]))) if seed % 2 else BvhHeuristic.Midpoint
    return objs, heur, scale, verts, idx


def grazing_rays(rng, verts, idx, scale, n):
    """Rays aimed at a point inside a chosen triangle, their direction in that triangle's own plane plus a
    component 10^-13 .. 10^-3 along its normal: Moeller-Trumbore's denominator P.e1 is then all cancellation."""
    v = verts.astype(np.float64)
    tri = idx[rng.integers(0, len(idx), size=n)]
    p1, p2, p3 = v[tri[:, 0]], v[tri[:, 1]], v[tri[:, 2]]
    nrm = np.cross(p2 - p1, p3 - p1)
    nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-300)
    b = rng.dirichlet((1.0, 1.0, 1.0), size=n)
    target = b[:, :1] * p1 + b[:, 1:2] * p2 + b[:, 2:] * p3
    inplane = (p2 - p1) * rng.normal(size=(n, 1)) + (p3 - p1) * rng.normal(size=(n, 1))
    inplane /= np.maximum(np.linalg.norm(inplane, axis=1, keepdims=True), 1e-300)
    eps = 10.0 ** rng.uniform(-13, -3, size=(n, 1))
    d = inplane + eps * rng.choice([-1.0, 1.0], size=(n, 1)) * nrm
    dist = scale * 10.0 ** rng.uniform(-3, 6, size=(n, 1))
    o = target - d * dist
    return o, d * 10.0 ** rng.uniform(-3, 3, size=(n, 1)), eps[:, 0]


def rays_for(rng, verts, scale, n):
    """Targets on the mesh (vertices, edge midpoints, random points of the bounding rectangle), directions grazing
    the sheet at angles 1e-7 .. 1 rad, origins 1e-3 .. 1e6 scene sizes back along the direction."""
    lo, hi = verts.min(axis=0).astype(np.float64), verts.max(axis=0).astype(np.float64)
    pick = rng.integers(0, len(verts), size=n)
    target = verts[pick].astype(np.float64)
    jitter = rng.uniform(-1, 1, size=(n, 3)) * (hi - lo + 1e-30) * (10.0 ** rng.uniform(-9, -1, size=(n, 1)))
    target = np.where(rng.random((n, 1)) < 0.5, target + jitter, rng.uniform(lo, hi, size=(n, 3)))
    ang = 10.0 ** rng.uniform(-7, 0, size=n)
    phi = rng.uniform(0, 2 * np.pi, size=n)
    d = np.stack([np.cos(phi) * np.cos(ang), np.sin(ang) * rng.choice([-1.0, 1.0], size=n), np.sin(phi) * np.cos(ang)], axis=1)
    axis_aligned = rng.random(n) < 0.05  # exact zeros in the direction: 0 * inf in the slab test
    d[axis_aligned] = np.eye(3)[rng.integers(0, 3, size=int(axis_aligned.sum()))] * rng.choice([-1.0, 1.0], size=(int(axis_aligned.sum()), 1))
    dist = scale * 10.0 ** rng.uniform(-3, 6, size=(n, 1))
    o = target - d * dist
    d = d * 10.0 ** rng.uniform(-3, 3, size=(n, 1))  # directions are not unit in the reference (lib.rs:202-210)
    return o, d


NEAR = 8.0        # abi.cpp RAYRS_FAR_DIAGONALS: diagonals of the root box between it and the origin
NEAR_PRIMS = 4096.0  # abi.cpp RAYRS_FAR_PRIMITIVES: ... and small-primitive sizes (5th percentile of the largest extents)


def small_extent(verts, idx):
    """FlatScene::small_extent for a scene of these triangles (scene_host.cpp: nth_element at n / 20)."""
    v = verts.astype(np.float64)[idx]
    ext = (v.max(axis=1) - v.min(axis=1)).max(axis=1)
    ext = np.where(ext > 0, ext, np.inf)
    return float(np.partition(ext, len(ext) // 20)[len(ext) // 20])


def is_near(o, root_box, small=0.0):
    """abi.cpp camera_is_far, negated: the origins from which a frame's camera gets the fast walk when it asks for it."""
    b = np.asarray(root_box, dtype=np.float64)
    lo, hi = b[0::2], b[1::2]
    out = np.maximum(np.maximum(lo - o, o - hi), 0.0)
    d2 = (out * out).sum(axis=1)
    near = d2 <= NEAR * NEAR * ((hi - lo) ** 2).sum()
    if small > 0.0 and np.isfinite(small):
        near &= d2 <= NEAR_PRIMS * NEAR_PRIMS * small * small
    return near


def families(seed, verts, idx, scale, per_scene, root_box, small=0.0):
    """[(family name, origins, directions)] for one scene."""
    rr = np.random.default_rng(seed * 7919 + 1)
    og, dg = rays_for(rr, verts, scale, per_scene // 2)
    oz, dz, eps = grazing_rays(rr, verts, idx, scale, per_scene - per_scene // 2)
    near = is_near(oz, root_box, small)
    out = [("general", og, dg)]
    for name, sel in (("grazing >= 1e-7", eps >= 1e-7), ("grazing 1e-9..1e-7", (eps < 1e-7) & (eps >= 1e-9)), ("grazing < 1e-9", eps < 1e-9)):
        out.append((name + " near", oz[sel & near], dz[sel & near]))
        out.append((name + " far", oz[sel & ~near], dz[sel & ~near]))
    return out


REQUIRED = ("general", "grazing >= 1e-7 near")

# A finely tessellated sheet of slivers, as a property of the geometry (not of the seed): at least
# FINE_TRIANGLES triangles whose median aspect ratio -- longest edge squared over twice the area -- is at least
# SLIVER_ASPECT.  Measured on forty sheets of 250 ... 400 quads per side: the fast walk's culling loses hits to in-plane
# rays at 1e-7 rad and more, from any distance, on sheets of aspect 2.7e4 and more (5 ... 17 of 24,000 near grazing rays
# on three of eight such sheets), on none of aspect 1.7e4 or less.
FINE_TRIANGLES = 100_000
SLIVER_ASPECT = 1.0e4


def fine_slivers(verts, idx):
    v = verts.astype(np.float64)[idx]
    e = np.stack([v[:, 1] - v[:, 0], v[:, 2] - v[:, 1], v[:, 0] - v[:, 2]], axis=1)
    longest2 = (e * e).sum(axis=2).max(axis=1)
    area2 = np.linalg.norm(np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0]), axis=1)
    aspect = float(np.median(longest2 / np.maximum(area2, 1e-300)))
    return len(idx) >= FINE_TRIANGLES and aspect >= SLIVER_ASPECT


def required(fine, name):
    """Where the fast walk is required to match: the general family everywhere, and near grazing rays at 1e-7 rad and more
    except on finely tessellated sliver sheets (fine_slivers) -- mismatches there are counted under their own heading."""
    return name in REQUIRED and not (fine and name != "general")


def main():
    millions = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    per_scene = 250_000
    n_scenes = max(1, int(millions * 1e6 / per_scene))
    hdri = np.zeros((2, 2, 3), dtype=np.float32)
    fam = {}
    t0 = time.time()
    exact_bad = 0
    hot_bad = 0
    n_hot_scenes = 0
    for seed in range(first, first + n_scenes):
        objs, heur, scale, verts, idx = scene_for(seed)
        tmin, tmax = 1e-6 * scale, 1e9 * scale
        prod = rayrs_amd.Scene(objs, tmin, tmax, heur, hdri, device=-1)
        fine = fine_slivers(verts, idx)
        osc = _oracle.OracleScene(objs, tmin, tmax, heur, hdri).use_walk_tree(prod)
        osg = _oracle.OracleScene(objs, tmin, tmax, heur, hdri).use_walk_tree(prod, gate=True)
        # a scene with a hot group (round 6: the floor fifty sheets wide under the sheet, seed % 13 == 0): the default walk as the
        # kernels make it there -- the tree without the group, the group tested beside it -- must match on every ray as well
        osh = _oracle.OracleScene(objs, tmin, tmax, heur, hdri).use_product_walk(prod, hot=True) if prod.info()["hot_count"] else None
        n_hot_scenes += osh is not None
        for name, o, d in families(seed, verts, idx, scale, per_scene, prod.info()["root_box"], small_extent(verts, idx)):
            f = fam.setdefault(name, dict(rays=0, hits=0, default=0, leaves=0, exact=0, worst=-1.0, in_front=0, beyond=0))
            ta, oa = osc.intersect_batch(o, d, tmin, tmax, traversal=0)
            tb, ob = osc.intersect_batch(o, d, tmin, tmax, traversal=2)
            try:
                _oracle.set_cull_margin(float("inf"))
                tl, ol = osc.intersect_batch(o, d, tmin, tmax, traversal=2)
                tx, ox = osg.intersect_batch(o, d, tmin, tmax, traversal=2)
                if osh is not None:
                    th, oh = osh.intersect_batch(o, d, tmin, tmax, traversal=2)
                    hot_bad += int(((oa != oh) | (ta.view(np.uint64) != th.view(np.uint64))).sum())
            finally:
                _oracle.set_cull_margin(2.0 ** -10)
            differs = lambda t, ob_: (oa != ob_) | (ta.view(np.uint64) != t.view(np.uint64))
            bad, badl, badx = differs(tb, ob), differs(tl, ol), differs(tx, ox)
            w, nf, nb = osc.cull_margin_probe(o, d, tmin, tmax)
            f["rays"] += len(o); f["hits"] += int((oa >= 0).sum())
            f["default"] += int(bad.sum()); f["leaves"] += int(badl.sum()); f["exact"] += int(badx.sum())
            f["required"] = f.get("required", 0) + (int(bad.sum()) if required(fine, name) else 0)
            f["exempt"] = f.get("exempt", 0) + (int(bad.sum()) if (name in REQUIRED and not required(fine, name)) else 0)
            f["worst"] = max(f["worst"], w); f["in_front"] += nf; f["beyond"] += nb
            exact_bad += int(badx.sum())
            for which, bb, tw, ow in (("default", bad, tb, ob), ("exact", badx, tx, ox)):
                if bb.any() and (which == "exact" or required(fine, name)):
                    i = int(np.argmax(bb))
                    print(f"MISMATCH ({which} walk, {name}) seed {seed}: o={o[i].tolist()} d={d[i].tolist()} reference=({oa[i]}, {ta[i]!r}) "
                          f"walk=({ow[i]}, {tw[i]!r})", flush=True)
        if (seed - first) % 16 == 15 or seed == first + n_scenes - 1:
            for name, f in fam.items():
                lw = np.log2(f["worst"]) if f["worst"] > 0 else float("-inf")
                print(f"{name:24s} {f['rays'] / 1e6:7.2f} M rays {f['hits'] / 1e6:7.2f} M hits  mismatches: default {f['default']:5d} (leaf boxes alone {f['leaves']:5d})  "
                      f"exact {f['exact']}  largest (entry - t)/t {f['worst']:.3e} (2^{lw:.1f})  hits in front of a box {f['in_front']}, beyond the margin {f['beyond']}", flush=True)
            print(f"  {time.time() - t0:.0f} s", flush=True)
    req = sum(f.get("required", 0) for f in fam.values())
    exempt = sum(f.get("exempt", 0) for f in fam.values())
    print("scenes with a hot group:", n_hot_scenes, "; the default walk with the group beside the tree, mismatches (must be 0):", hot_bad)
    exact_bad += hot_bad
    print("done:", sum(f["rays"] for f in fam.values()), "rays; default (exact) walk mismatches (must be 0):", exact_bad,
          "; fast walk mismatches where it must match:", req,
          "; on finely tessellated sliver sheets, in families it must match elsewhere (counted, not required):", exempt,
          "; in the other families:", sum(f["default"] for f in fam.values()) - req - exempt)
    sys.exit(1 if (req or exact_bad) else 0)


if __name__ == "__main__":
    main()
