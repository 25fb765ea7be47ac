"""The two routes of the radiance integrator render the same bits.

Scenes whose gate tree is at most one record (the reference's example scenes: a floor and one to
seven spheres, test_scenes.rs:14-256) are rendered by ONE launch that keeps every path in LDS from its
first ray to its last (rayrs_amd/csrc/local_pool.hip); everything else -- and the same scenes with
rayrs_tuning.local_pool = 1 -- streams its paths through the pool in HBM, three launches per bounce
(wavefront.hip).  Both call device_path.h's functions, so both must reproduce the oracle's frame bit
for bit, its ray / path / escaped-path counts, and the work counters of the oracle's walk on the tree each
walks (the local pool: the groups behind their gating boxes; the streaming kernels: the default tree of single
primitives behind their own boxes)."""
import numpy as np
import pytest

import _oracle
import rayrs_amd
from rayrs_amd import procedural, scenes
from rayrs_amd.api import BvhHeuristic, Emission, Material, Object, Fresnel, Axis

pytestmark = pytest.mark.gpu

HDRI = procedural.make_hdri(256, 128)


def frames(cam_args, objs, heur, w, h, spp, mb, chunk=0, seed=0x5EED, count_work=False):
    cam_args = scenes.camera_for_resolution(cam_args, w, h)
    scene = rayrs_amd.Scene(objs, 1e-6, 1e6, heur, HDRI, device=0)
    cam = rayrs_amd.Camera(*cam_args)
    assert scene.info()["local_pool"] == 1
    loc, lst = rayrs_amd.render(scene, cam, spp, mb, seed=seed, sample_chunk=chunk, out_f64=True, count_work=count_work)
    scene.set_tuning(local_pool=1)
    assert scene.info()["local_pool"] == 0
    stream, sst = rayrs_amd.render(scene, cam, spp, mb, seed=seed, sample_chunk=chunk, out_f64=True,
                                   count_work=count_work)
    osc = _oracle.OracleScene(objs, 1e-6, 1e6, heur, HDRI)
    ocam = _oracle.OracleCamera(*cam_args)
    ref, ost = osc.render(ocam, spp, mb, seed=seed, sample_chunk=chunk, traversal=0)
    if count_work:  # the oracle's walk on the records each route walks
        _, ost["gate_walk"] = osc.use_walk_tree(scene, gate=True).render(ocam, spp, mb, seed=seed, sample_chunk=chunk, traversal=2)
        _, ost["default_walk"] = osc.use_product_walk(scene).render(ocam, spp, mb, seed=seed, sample_chunk=chunk, traversal=2)
    return (loc, lst), (stream, sst), (ref, ost)


def same_bits(a, b):
    return np.array_equal(a.view(np.uint64), b.view(np.uint64))


CASES = [
    ("diffuse_single_sphere", scenes.diffuse_single_sphere, 64, 48, 16, 50, 0),     # no record at all: the root group
    ("copper_single_sphere", scenes.copper_single_sphere, 48, 32, 8, 50, 0),
    ("glass_single_sphere", scenes.glass_single_sphere, 48, 32, 16, 50, 4),
    ("cook_torrance_glass_single_sphere", scenes.cook_torrance_glass_single_sphere, 48, 32, 16, 50, 0),
    ("spheres_metallic", scenes.cook_torrance_spheres_metallic, 96, 40, 16, 50, 4),  # one record, three gates
    ("spheres_plastic", scenes.cook_torrance_spheres_plastic, 96, 40, 8, 50, 0),
    ("spheres_frosted_glass", scenes.cook_torrance_spheres_frosted_glass, 96, 40, 16, 32, 4),
    ("spheres_ct_refract", scenes.cook_torrance_spheres_cook_torrance_refract, 96, 40, 8, 50, 0),
    ("material_test", scenes.material_test, 128, 24, 16, 50, 5),                     # eight material kinds in one wave
    ("ragged_image_edge", scenes.material_test, 61, 19, 4, 50, 0),
    ("bounce_budget_3", scenes.cook_torrance_spheres_frosted_glass, 64, 32, 8, 3, 0),
    ("bounce_budget_1", scenes.cook_torrance_spheres_metallic, 64, 32, 4, 1, 0),
]


@pytest.mark.parametrize("name,scene_fn,w,h,spp,mb,chunk", CASES, ids=[c[0] for c in CASES])
def test_local_pool_streaming_and_oracle_agree(name, scene_fn, w, h, spp, mb, chunk):
    cam_args, objs, heur = scene_fn()
    (loc, lst), (stream, sst), (ref, ost) = frames(cam_args, objs, heur, w, h, spp, mb, chunk, count_work=True)
    for st in (lst, sst):
        assert st["rays"] == ost["rays"] and st["paths"] == ost["paths"] == w * h * spp
        assert st["escaped_paths"] == ost["escaped_paths"]
        assert st["nan_pixels"] == ost["nan_pixels"] and st["neg_pixels"] == ost["neg_pixels"]
    assert lst["kernel_launches"] == 1 and sst["kernel_launches"] > 1
    assert same_bits(loc, ref), f"local pool: {int((loc != ref).any(axis=2).sum())} pixels differ from the oracle"
    assert same_bits(stream, ref)
    # the work of each route: records entered and primitive tests by kind are the oracle's on the tree it walks;
    # hits per surface and direct rays do not depend on the tree
    for k in ("interior_visits", "tri_tests", "sphere_tests", "plane_tests"):
        assert lst[k] == ost["gate_walk"][k], k
        assert sst[k] == ost["default_walk"][k], k
    for k in ("surface_hits", "direct_rays"):
        assert lst[k] == sst[k], k


def test_emitters_and_light_side_array():
    """Paths that have met an emitter carry `light` (lib.rs:534); the local pool keeps it in a side array in HBM,
    as the streaming route does.  Emitters of several material kinds, met mid-path."""
    floor = Object.plane(Axis.Y, -25., 25., -25., 25., 0.,
                         Material.CookTorrance((1., 1., 1.), 0.5, Fresnel.SchlickMetallic((0.8, 0.8, 0.8))),
                         Emission.Dark())
    objs = [floor,
            Object.sphere(1., (-2.5, 1., 0.), Material.LambertianDiffuse((0.8, 0.5, 0.2)), Emission.Emissive(2.0, (1., 0.9, 0.8))),
            Object.sphere(1., (0., 1., 0.), Material.Glass((1., 1., 1.), 1.5), Emission.Emissive(0.5, (0.2, 0.4, 1.))),
            Object.sphere(1., (2.5, 1., 0.), Material.NoReflect(), Emission.Emissive(5.0, (1., 1., 1.))),  # lib.rs:550
            Object.sphere(0.5, (0., 0.5, 2.), Material.Reflect((0.9, 0.9, 0.9)), Emission.Dark())]
    cam_args = ((0., 4., 9.), (0., 1., 0.), (0., 1., 0.), 50., 4., 2., 100)
    (loc, lst), (stream, sst), (ref, ost) = frames(cam_args, objs, BvhHeuristic.Sah(1000), 80, 40, 16, 50, 4)
    assert lst["rays"] == sst["rays"] == ost["rays"]
    assert same_bits(loc, ref) and same_bits(stream, ref)
    assert ref.max() > 1.0


def test_horizon_and_sky_primary_rays():
    """Primary rays that miss the root Node's box (bvh.rs:394) never enter ISECT: they count as queries and as
    escaped paths, exactly as on the streaming route (direct_rays)."""
    cam_args, objs, heur = scenes.diffuse_single_sphere()
    cam_args = ((0., 5., 10.), (0., 1., 0.), (0., 5.5, 0.), 70., 4., 3., 100)  # the horizon through the frame
    (loc, lst), (stream, sst), (ref, ost) = frames(cam_args, objs, heur, 64, 48, 8, 50, 0)
    assert lst["direct_rays"] == sst["direct_rays"] > 0
    assert lst["rays"] == sst["rays"] == ost["rays"] and lst["escaped_paths"] == ost["escaped_paths"]
    assert same_bits(loc, ref) and same_bits(stream, ref)


def test_tile_shares_sum_to_the_frame():
    cam_args, objs, heur = scenes.cook_torrance_spheres_metallic()
    cam_args = scenes.camera_for_resolution(cam_args, 100, 44)  # ragged: edge tiles with padding pixels
    scene = rayrs_amd.Scene(objs, 1e-6, 1e6, heur, HDRI, device=0)
    cam = rayrs_amd.Camera(*cam_args)
    full, st = rayrs_amd.render(scene, cam, 8, 50, seed=3, sample_chunk=4)
    parts = [rayrs_amd.render(scene, cam, 8, 50, seed=3, sample_chunk=4, tile_rank=r, tile_ranks=3) for r in range(3)]
    assert np.array_equal(parts[0][0] + parts[1][0] + parts[2][0], full)
    assert sum(p[1]["rays"] for p in parts) == st["rays"]


def test_selection_rule_on_both_sides_of_the_threshold():
    """One record (or none) -> the local pool; a second record -> the streaming kernels, whatever the tuning says."""
    rng = np.random.default_rng(11)
    mat = Material.LambertianDiffuse((0.7, 0.7, 0.7))

    def spheres(n):
        return [Object.sphere(0.4, (float(x), 0.4, float(z)), mat, Emission.Dark())
                for x, z in rng.uniform(-4, 4, size=(n, 2))]
    floor = Object.plane(Axis.Y, -25., 25., -25., 25., 0., mat, Emission.Dark())
    cam_args = scenes.camera_for_resolution(((0., 5., 10.), (0., 1., 0.), (0., 0.5, 0.), 50., 4., 3., 100), 48, 36)
    seen = set()
    for n in (1, 3, 6, 9, 14, 20, 40):
        objs = [floor] + spheres(n)
        scene = rayrs_amd.Scene(objs, 1e-6, 1e6, BvhHeuristic.Sah(1000), HDRI, device=0)
        info = scene.info()
        assert info["local_pool"] == (1 if info["gate_n_wide"] <= 1 and info["n_prims"] <= 16 else 0)
        seen.add(info["local_pool"])
        cam = rayrs_amd.Camera(*cam_args)
        img, st = rayrs_amd.render(scene, cam, 4, 50, seed=5, out_f64=True)
        assert (st["kernel_launches"] == 1) == (info["local_pool"] == 1)
        osc = _oracle.OracleScene(objs, 1e-6, 1e6, BvhHeuristic.Sah(1000), HDRI)
        ref, ost = osc.render(_oracle.OracleCamera(*cam_args), 4, 50, seed=5, traversal=0)
        assert st["rays"] == ost["rays"] and same_bits(img, ref)
    assert seen == {0, 1}


def test_config1_whole_frame_at_its_stated_size():
    """BASELINE.json configs[0] in full: diffuse_single_sphere, 256 x 256, 64 spp (test_scenes.rs:14-44) -- small
    enough for the oracle to render every pixel, so the whole frame is compared, not a band."""
    cam_args, objs, heur, spp, mb = scenes.config(1)
    scene = rayrs_amd.Scene(objs, 1e-6, 1e6, heur, HDRI, device=0)
    cam = rayrs_amd.Camera(*cam_args)
    assert (cam.x_pixels(), cam.y_pixels(), spp, mb) == (256, 256, 64, 50)
    chunk = rayrs_amd.frame_sample_chunk(256, 256, spp)
    osc = _oracle.OracleScene(objs, 1e-6, 1e6, heur, HDRI)
    ref, ost = osc.render(_oracle.OracleCamera(*cam_args), spp, mb, seed=0x5EED, sample_chunk=chunk, traversal=0)
    for local_pool in (0, 1):
        scene.set_tuning(local_pool=local_pool)
        img, st = rayrs_amd.render(scene, cam, spp, mb, seed=0x5EED, sample_chunk=chunk, out_f64=True)
        assert st["paths"] == 256 * 256 * 64 == ost["paths"] and st["rays"] == ost["rays"]
        assert same_bits(img, ref)


def test_segments_of_whole_tiles_are_resolved_behind_their_launch():
    """The local-pool route renders a frame as launches over segments of whole tiles (2^27 items by default: config 4's
    2^30 items are eight) and resolves each segment behind its launch, so the item-sum array holds one segment
    (3.2 GB instead of 25.8 on config 4).  Here: segments of 65536 items on a ragged frame -- seven of them, the
    last one short -- against one segment and the oracle."""
    cam_args, objs, heur = scenes.cook_torrance_spheres_frosted_glass()
    cam_args = scenes.camera_for_resolution(cam_args, 203, 117)   # 26 x 15 tiles, padding pixels on two edges
    scene = rayrs_amd.Scene(objs, 1e-6, 1e6, heur, HDRI, device=0)
    cam = rayrs_amd.Camera(*cam_args)
    one, st1 = rayrs_amd.render(scene, cam, 64, 32, seed=9, sample_chunk=4, out_f64=True)
    scene.lab_set(local_segment_items=65536)
    many, stn = rayrs_amd.render(scene, cam, 64, 32, seed=9, sample_chunk=4, out_f64=True)
    assert st1["kernel_launches"] == 1 and stn["kernel_launches"] == 7   # 390 tiles x 16 chunks x 64 = 399360 items
    assert stn["rays"] == st1["rays"] and stn["paths"] == 203 * 117 * 64
    assert same_bits(one, many)
    osc = _oracle.OracleScene(objs, 1e-6, 1e6, heur, HDRI)
    ref, ost = osc.render(_oracle.OracleCamera(*cam_args), 64, 32, seed=9, sample_chunk=4, traversal=0)
    assert same_bits(many, ref) and stn["rays"] == ost["rays"]


def test_a_scene_at_the_local_pools_limits_renders_with_fewer_workgroups_per_cu():
    """Sixteen primitives in four groups of four and sixteen distinct surface rows: the most a local-pool scene may hold
    (LP_MAX_PRIMS).  Its primitive records and surface rows take so much of a workgroup's LDS that fewer than the three
    workgroups per CU the kernel is built for fit (the launch takes the count from the occupancy query
    with the scene's real LDS size -- if it assumed three, the launch would fail or the frame be wrong).  Behavioural: the
    frame, ray and path counts are the oracle's on both routes."""
    from rayrs_amd.api import BvhHeuristic, Emission, Fresnel, Material, Object
    mats = [Material.LambertianDiffuse((0.8, 0.7, 0.6)), Material.Reflect((0.8, 0.8, 0.8)), Material.Glass((1, 1, 1), 1.45),
            Material.CookTorrance((1, 1, 1), 0.2, Fresnel.SchlickMetallic((0.8, 0.6, 0.4))),
            Material.Plastic((0.6, 0.7, 0.8), (1, 1, 1), 0.1, 1.45), Material.CookTorranceGlass((1, 1, 1), 0.1, 1.45),
            Material.Refract((1, 1, 1), 1.3), Material.CookTorranceRefract((1, 1, 1), 0.15, 1.45)]
    objs = []
    for i in range(16):
        m = mats[i % 8]
        e = Emission.new(2.0 + i, (1.0, 0.9, 0.8)) if i in (3, 12) else Emission.Dark()
        # sixteen distinct rows: the same material with another emission, or another colour, is another row
        if i >= 8 and not e.emissive:
            e = Emission.new(0.0, (0.1 * (i - 7), 0.5, 0.5))
        objs.append(Object.sphere(0.45, (1.1 * (i - 7.5), 1.0, 0.3 * ((i * 7) % 3)), m, e))
    heur = BvhHeuristic.Midpoint
    cam_args = ((0.0, 6.0, 16.0), (0.0, 1.0, 0.0), (0.0, 1.0, 0.0), 65.0, 96 / 254.0, 48 / 254.0, 100)
    scene = rayrs_amd.Scene(objs, 1e-6, 1e6, heur, HDRI, device=0)
    info = scene.info()
    assert info["local_pool"] == 1 and info["n_prims"] == 16 and info["n_surfaces"] == 16 and info["gate_n_wide"] == 1
    cam = rayrs_amd.Camera(*cam_args)
    osc = _oracle.OracleScene(objs, 1e-6, 1e6, heur, HDRI)
    ocam = _oracle.OracleCamera(*cam_args)
    ref, ost = osc.render(ocam, 24, 50, seed=11, sample_chunk=4, traversal=0)
    loc, lst = rayrs_amd.render(scene, cam, 24, 50, seed=11, sample_chunk=4, out_f64=True, count_work=True)
    assert lst["local_pool"] == 1 and lst["rays"] == ost["rays"] and lst["paths"] == ost["paths"]
    assert same_bits(loc, ref)
    _, wst = osc.use_product_walk(scene).render(ocam, 24, 50, seed=11, sample_chunk=4, traversal=2)
    for k in ("interior_visits", "sphere_tests"):
        assert lst[k] == wst[k], k
    scene.set_tuning(local_pool=1)
    stream, sst = rayrs_amd.render(scene, cam, 24, 50, seed=11, sample_chunk=4, out_f64=True)
    assert sst["local_pool"] == 0 and sst["rays"] == ost["rays"] and same_bits(stream, ref)
