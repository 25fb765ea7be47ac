"""GPU parity of the whole hot path: the frame rendered by the gfx950 kernel
through the C ABI against the CPU oracle's frame, same scene, same seed.

Bar: the f64 framebuffer is bit-identical to the oracle's (which is far inside
the north star's 1e-4 relative L2); ray counts are equal; the f32 framebuffer is
the f64 one rounded once (image.rs:224-229)."""
import numpy as np
import pytest

import _oracle
import rayrs_amd
from rayrs_amd import procedural, scenes

pytestmark = pytest.mark.gpu

HDRI = procedural.make_hdri(256, 128)


def both(scene_fn, w, h, spp, max_bounces=50, seed=0x5EED, **kw):
    cam_args, objs, heur = scene_fn()
    cam_args = scenes.camera_for_resolution(cam_args, w, h)
    scene = rayrs_amd.Scene(objs, 1e-6, 1e6, heur, HDRI, device=0)
    cam = rayrs_amd.Camera(*cam_args)
    osc = _oracle.OracleScene(objs, 1e-6, 1e6, heur, HDRI)
    ocam = _oracle.OracleCamera(*cam_args)
    assert (cam.x_pixels(), cam.y_pixels()) == (w, h) == (ocam.x_pixels(), ocam.y_pixels())
    return scene, cam, osc, ocam


def assert_same_frame(img, ref):
    a, b = img.view(np.uint64), ref.view(np.uint64)
    if not np.array_equal(a, b):
        bad = (a != b).any(axis=2)
        num = np.sqrt(((img - ref) ** 2).sum())
        den = np.sqrt((ref ** 2).sum())
        raise AssertionError(f"{int(bad.sum())} of {bad.size} pixels differ; relative L2 = {num / den:.3e}; "
                             f"first at {np.argwhere(bad)[0]}")


CASES = [
    # name, scene, W, H, spp, max_bounces
    ("diffuse_single_sphere", scenes.diffuse_single_sphere, 64, 48, 16, 50),          # config 1 family
    ("copper_single_sphere", scenes.copper_single_sphere, 48, 32, 8, 50),
    ("glass_single_sphere", scenes.glass_single_sphere, 48, 32, 16, 50),
    ("cook_torrance_glass_single_sphere", scenes.cook_torrance_glass_single_sphere, 48, 32, 16, 50),
    ("spheres_metallic", scenes.cook_torrance_spheres_metallic, 96, 40, 16, 50),      # config 2 family
    ("spheres_plastic", scenes.cook_torrance_spheres_plastic, 96, 40, 8, 50),
    ("spheres_frosted_glass", scenes.cook_torrance_spheres_frosted_glass, 96, 40, 16, 32),  # config 4 family
    ("spheres_ct_refract", scenes.cook_torrance_spheres_cook_torrance_refract, 96, 40, 8, 50),
    ("material_test", scenes.material_test, 128, 24, 16, 50),
    ("mesh_1280_light", lambda: scenes.mesh_scene(3, rayrs_amd.Material.LambertianDiffuse((0.8, 0.8, 0.8)),
                                                  area_light=True), 64, 48, 8, 50),  # config 3 family
    ("mesh_5120_copper", lambda: scenes.mesh_scene(4), 64, 48, 4, 50),                # config 5 family
    ("ragged_image_edge", scenes.material_test, 61, 19, 4, 50),                       # W, H not multiples of 8
    ("bounce_budget_3", scenes.cook_torrance_spheres_frosted_glass, 64, 32, 8, 3),    # lib.rs:559
]


@pytest.mark.parametrize("name,scene_fn,w,h,spp,mb", CASES, ids=[c[0] for c in CASES])
def test_frame_bit_identical_to_oracle(name, scene_fn, w, h, spp, mb):
    scene, cam, osc, ocam = both(scene_fn, w, h, spp, mb)
    img, st = rayrs_amd.render(scene, cam, spp, mb, seed=0x5EED, out_f64=True)
    ref, ost = osc.render(ocam, spp, mb, seed=0x5EED, traversal=0)
    assert st["rays"] == ost["rays"] and st["paths"] == ost["paths"] == w * h * spp
    assert st["nan_pixels"] == ost["nan_pixels"] and st["neg_pixels"] == ost["neg_pixels"]
    assert_same_frame(img, ref)


def test_f32_framebuffer_is_rounded_f64():
    scene, cam, osc, ocam = both(scenes.material_test, 64, 16, 8)
    img64, _ = rayrs_amd.render(scene, cam, 8, out_f64=True)
    img32, _ = rayrs_amd.render(scene, cam, 8, out_f64=False)
    assert img32.dtype == np.float32
    assert np.array_equal(img32, img64.astype(np.float32))


def test_chunked_sum_matches_oracle_and_sequential():
    scene, cam, osc, ocam = both(scenes.cook_torrance_spheres_metallic, 64, 32, 24)
    seq, _ = rayrs_amd.render(scene, cam, 24, out_f64=True)
    chk, st = rayrs_amd.render(scene, cam, 24, sample_chunk=5, out_f64=True)  # 5 chunks, last one short
    ref, ost = osc.render(ocam, 24, sample_chunk=5)
    assert st["rays"] == ost["rays"]
    assert_same_frame(chk, ref)
    assert np.allclose(chk, seq, rtol=1e-12, atol=1e-15)


def test_work_counters_match_oracle_ordered_traversal():
    """The kernel's traversal counters (count_work) equal the oracle's instrumented
    ordered traversal: same visit order, so the algorithmic-bytes figure of the
    roofline line can be computed on either side."""
    scene, cam, osc, ocam = both(lambda: scenes.mesh_scene(3, area_light=True), 48, 32, 4)
    for fast in (False, True):  # the default walk, and rayrs_render_params.fast_traversal
        img, st = rayrs_amd.render(scene, cam, 4, out_f64=True, count_work=True, fast_traversal=fast)
        ref, ost = osc.use_product_walk(scene, fast=fast).render(ocam, 4, traversal=2)  # the kernel's walk on the kernel's records
        assert_same_frame(img, ref)
        for k in ("rays", "interior_visits", "tri_tests", "sphere_tests", "plane_tests", "escaped_paths"):
            assert st[k] == ost[k], (fast, k)


def test_traversal_stack_overflow_strip():
    """The traversal keeps the first entries of a lane's stack in LDS and the rest in an HBM
    strip (device_path.h LaneStack).  With only two entries in LDS nearly every query uses the
    strip; the frame and the work counters must not change."""
    scene, cam, osc, ocam = both(lambda: scenes.mesh_scene(4), 64, 48, 4)
    scene.lab_set(stack_lds=2)
    assert scene.info()["wide_depth"] > 8 and scene.info()["gate_depth"] > 8
    for fast in (False, True):
        img, st = rayrs_amd.render(scene, cam, 4, out_f64=True, count_work=True, fast_traversal=fast)
        ref, ost = osc.use_product_walk(scene, fast=fast).render(ocam, 4, traversal=2)
        assert_same_frame(img, ref)
        for k in ("rays", "interior_visits", "tri_tests", "plane_tests"):
            assert st[k] == ost[k], (fast, k)


def test_deep_chain_tree_beyond_lds():
    """A tree that peels two objects per level (pairs of spheres at 1.5**k under the midpoint
    heuristic; one-object sides would trigger the median fallback, bvh.rs:279-287) is 125 levels
    deep: ten times what the traversal keeps in LDS and beyond any fixed-size stack.  The
    reference recurses without a limit; so must this."""
    from rayrs_amd.api import BvhHeuristic, Emission, Material, Object
    objs = []
    for k in range(250):
        x = 1.5 ** k
        for j in range(2):
            objs.append(Object.sphere(0.25 * x if k > 3 else 0.2, (x * (1 + 0.01 * j), 1.0, 0.0),
                                      Material.LambertianDiffuse((0.8, 0.8, 0.8)), Emission.Dark()))
    cam_args = ((-3.0, 1.5, 4.0), (0.0, 1.0, 0.0), (20.0, 1.0, 0.0), 70.0, 64 / 254.0, 32 / 254.0, 100)
    scene = rayrs_amd.Scene(objs, 1e-6, 1e60, BvhHeuristic.Midpoint, HDRI, device=0)
    assert scene.info()["wide_depth"] > 100 and scene.info()["gate_depth"] > 100
    cam = rayrs_amd.Camera(*cam_args)
    osc = _oracle.OracleScene(objs, 1e-6, 1e60, BvhHeuristic.Midpoint, HDRI)
    ocam = _oracle.OracleCamera(*cam_args)
    img, st = rayrs_amd.render(scene, cam, 4, out_f64=True, count_work=True)
    ref, ost = osc.render(ocam, 4, traversal=0)
    assert_same_frame(img, ref)
    assert st["rays"] == ost["rays"] and st["sphere_tests"] > st["rays"]
    ref2, ost2 = osc.use_product_walk(scene).render(ocam, 4, traversal=2)
    assert st["interior_visits"] == ost2["interior_visits"] and st["sphere_tests"] == ost2["sphere_tests"]
    img, st = rayrs_amd.render(scene, cam, 4, out_f64=True, count_work=True, fast_traversal=True)
    assert_same_frame(img, ref)
    ref2, ost2 = osc.use_product_walk(scene, fast=True).render(ocam, 4, traversal=2)
    assert st["interior_visits"] == ost2["interior_visits"] and st["sphere_tests"] == ost2["sphere_tests"]


def test_launch_limits_are_reported_not_rendered():
    """What the path pool cannot number is refused with RAYRS_UNSUPPORTED (-5), never rendered
    wrongly: an image side beyond 16 bits (TailSlot::pix), more than 2^32 (pixel, chunk) items,
    a bounce budget beyond the pool's 16-bit bounce/draw counters, more samples per pixel than a slot's cursor counts."""
    from rayrs_amd import _ffi
    cam_args, objs, heur = scenes.diffuse_single_sphere()
    scene = rayrs_amd.Scene(objs, 1e-6, 1e6, heur, HDRI, device=0)

    def status(cam, **kw):
        try:
            rayrs_amd.render_launch(scene, cam, rayrs_amd.make_params(**kw), 16, 0)  # refused before the buffer is touched
        except _ffi.RayrsError as e:
            return e.status
        raise AssertionError("launch accepted")

    wide = rayrs_amd.Camera(*scenes.camera_for_resolution(cam_args, 70000, 8))
    assert status(wide, spp=1, max_bounces=4) == -5
    big = rayrs_amd.Camera(*scenes.camera_for_resolution(cam_args, 8192, 8192))
    assert status(big, spp=4096, max_bounces=4, sample_chunk=8) == -5  # 2^26 pixels * 512 chunks = 2^35 items
    small = rayrs_amd.Camera(*scenes.camera_for_resolution(cam_args, 8, 8))
    assert status(small, spp=1, max_bounces=9000) == -5
    assert status(small, spp=1, max_bounces=8001) == -5
    img, st = rayrs_amd.render(scene, small, 1, 8000)    # the limit itself renders
    assert st["paths"] == 64
    assert status(small, spp=(1 << 30) + 5, max_bounces=4) == -5          # a slot's sample cursor has 30 bits


def test_tile_sharding_is_exact():
    """Two 'ranks' rendering interleaved 8x8 tiles into zeroed buffers sum to the
    single-GPU frame exactly (x + 0): the multi-GPU reduce is order independent.  The pixels a
    rank writes are exactly rayrs_amd.tiles.tile_mask(rank) -- on a ragged image too."""
    from rayrs_amd import tiles
    scene, cam, osc, ocam = both(scenes.material_test, 93, 21, 6)
    full, st = rayrs_amd.render(scene, cam, 6, out_f64=False)
    parts = []
    rays = 0
    for r in range(3):
        canvas = np.full((21, 93, 3), -7.0, dtype=np.float32)  # untouched pixels keep the caller's values
        p, s = rayrs_amd.render(scene, cam, 6, tile_rank=r, tile_ranks=3, out_f64=False, out=canvas)
        assert np.array_equal((p != -7.0).all(axis=2), tiles.tile_mask(93, 21, r, 3))
        assert np.array_equal((p != -7.0).any(axis=2), tiles.tile_mask(93, 21, r, 3))
        parts.append(np.where(p == -7.0, np.float32(0), p))
        rays += s["rays"]
    assert rays == st["rays"]
    # disjoint support
    nz = [(p != 0).any(axis=2) for p in parts]
    assert not (nz[0] & nz[1]).any() and not (nz[1] & nz[2]).any() and not (nz[0] & nz[2]).any()
    assert np.array_equal(parts[0] + parts[1] + parts[2], full)


def test_seed_changes_image_and_same_seed_repeats():
    scene, cam, osc, ocam = both(scenes.diffuse_single_sphere, 32, 32, 4)
    a, _ = rayrs_amd.render(scene, cam, 4, seed=1, out_f64=True)
    b, _ = rayrs_amd.render(scene, cam, 4, seed=1, out_f64=True)
    c, _ = rayrs_amd.render(scene, cam, 4, seed=2, out_f64=True)
    assert np.array_equal(a, b)
    assert not np.array_equal(a, c)


@pytest.mark.parametrize("n,lit", [(2, True), (3, True), (2, False)])
def test_render_multi_on_logical_ranks_equals_the_single_device_frame(n, lit):
    """rayrs_render_multi (the block loop over several GPUs inside the library): n handles of the same
    scene, here all on device 0, one host thread and stream each, tiles t % n == rank, the buffers
    summed -- ranks sharing a device by the accumulate kernel, the distinct devices by one RCCL
    reduce (a one-rank communicator here).  The frame must equal the single-handle frame bit for bit."""
    # with the area light paths carry light in the side array (eager kernels), without it none does (wavefront.h)
    scene, cam, osc, ocam = both(lambda: scenes.mesh_scene(3, area_light=lit), 77, 45, 6)
    full, st = rayrs_amd.render(scene, cam, 6, sample_chunk=4, out_f64=False)
    clones = [scene] + [scene.clone_to_device(0) for _ in range(n - 1)]
    img, mst = rayrs_amd.render_multi(clones, cam, 6, sample_chunk=4)
    assert np.array_equal(img, full)
    assert mst["rays"] == st["rays"] and mst["paths"] == st["paths"] == 77 * 45 * 6
    img64, _ = rayrs_amd.render_multi(clones, cam, 6, sample_chunk=4, out_f64=True)
    ref, _ = osc.render(ocam, 6, sample_chunk=4)
    assert_same_frame(img64, ref)


@pytest.mark.parametrize("route", ["streaming", "local_pool"])
def test_render_multi_runs_its_rccl_reduce_on_one_device(route):
    """What a multi-GPU node executes after the ranks have rendered -- dlopen of librccl, ncclCommInitAll, the grouped
    in-place ncclReduce to rank 0 -- is skipped when every handle sits on one device (there is nothing to reduce).
    rayrs_lab.h force_rccl runs it anyway, with a communicator of that one device: the frame must not change by a bit,
    with one handle, with three and with the eight of a full node (summed on the device first), on either route, and
    librccl must really be mapped."""
    fn = (lambda: scenes.mesh_scene(3, area_light=True)) if route == "streaming" else scenes.cook_torrance_spheres_metallic
    scene, cam, osc, ocam = both(fn, 77, 45, 6)
    assert scene.info()["local_pool"] == (1 if route == "local_pool" else 0)
    full, st = rayrs_amd.render(scene, cam, 6, sample_chunk=4)
    scene.lab_set(force_rccl=1)
    clones = [scene] + [scene.clone_to_device(0) for _ in range(7)]   # clones take the scene's settings along
    for handles in (clones[:1], clones[:3], clones):                  # one, three, and a full node's eight ranks
        img, mst = rayrs_amd.render_multi(handles, cam, 6, sample_chunk=4)
        assert np.array_equal(img.view(np.uint32), full.view(np.uint32)), len(handles)
        assert mst["rays"] == st["rays"] and mst["paths"] == st["paths"] == 77 * 45 * 6
        assert mst["local_pool"] == st["local_pool"]
    assert any("librccl" in line for line in open("/proc/self/maps")), "RCCL was never loaded"
    img64, _ = rayrs_amd.render_multi(clones, cam, 6, sample_chunk=4, out_f64=True)   # ncclFloat64
    ref, _ = osc.render(ocam, 6, sample_chunk=4)
    assert_same_frame(img64, ref)


def test_the_default_walk_is_the_references_visit_set_and_the_fast_walk_visits_less():
    """The default walk: the reference's leaf groups behind their gating boxes, nothing culled by the closest hit so
    far -- the reference's own visit set (bvh.rs:391-415), equal to the oracle's walk of the gate tree with its margin
    set to infinity.  rayrs_render_params.fast_traversal renders the same frame visiting fewer records and testing
    fewer primitives (its two bets)."""
    scene, cam, osc, ocam = both(lambda: scenes.mesh_scene(4), 96, 64, 4)
    a, sa = rayrs_amd.render(scene, cam, 4, out_f64=True, count_work=True, fast_traversal=True)
    b, sb = rayrs_amd.render(scene, cam, 4, out_f64=True, count_work=True)
    ref, ost = osc.render(ocam, 4, traversal=0)
    assert_same_frame(a, ref)
    assert_same_frame(b, ref)
    assert sa["exact_walk"] == 0 and sb["exact_walk"] == 1
    assert sa["rays"] == sb["rays"] == ost["rays"]
    assert sb["interior_visits"] > sa["interior_visits"] and sb["tri_tests"] > sa["tri_tests"]
    _, wst = osc.use_product_walk(scene).render(ocam, 4, traversal=2)
    for k in ("interior_visits", "tri_tests", "plane_tests"):
        assert sb[k] == wst[k], k


def test_a_camera_far_from_the_scene_gets_the_default_walk_whatever_was_asked():
    """The fast walk's leaf boxes fail, a few times in 10^4, for rays aimed along a triangle's plane from thousands of
    primitive sizes away (profiles/r04_tight_leaves.txt) and were never seen to fail from nearby.  Bounce rays start on
    the scene; only a camera can stand far out -- and a frame whose camera is more than 8 root-box diagonals, or 4096 of
    the scene's small primitives, from the root box takes the default walk also when the fast one is asked for
    (abi.cpp camera_is_far).  rayrs_render_stats.exact_walk reports the walk; the far frame's counters are the oracle's
    for the gate tree with nothing culled."""
    mesh = lambda: scenes.mesh_scene(3)
    scene, cam, osc, ocam = both(mesh, 64, 48, 4)
    box = np.array(scene.info()["root_box"])
    diag = float(np.linalg.norm(box[1::2] - box[0::2]))
    _, st = rayrs_amd.render(scene, cam, 4, out_f64=True, fast_traversal=True)
    assert st["exact_walk"] == 0 and st["local_pool"] == 0           # the reference's own camera: the fast walk as asked
    _, st = rayrs_amd.render(scene, cam, 4, out_f64=True)
    assert st["exact_walk"] == 1
    # the same mesh through a long lens from 12 diagonals above the floor's corner
    far_o = (box[1] + 12.0 * diag * 0.6, 12.0 * diag * 0.8, 0.3)
    far = (far_o, (0.0, 1.0, 0.0), (0.0, 1.0, 0.0), 0.45, 64 / 254.0, 48 / 254.0, 100)
    near_o = tuple(0.5 * c for c in far_o)                             # 6 diagonals (and 3000 primitive sizes): the fast walk
    near = (near_o, (0.0, 1.0, 0.0), (0.0, 1.0, 0.0), 0.9, 64 / 254.0, 48 / 254.0, 100)
    for cam_args, want in ((far, 1), (near, 0)):
        cam = rayrs_amd.Camera(*cam_args)
        ocam = _oracle.OracleCamera(*cam_args)
        img, st = rayrs_amd.render(scene, cam, 6, out_f64=True, count_work=True, fast_traversal=True)
        ref, ost = osc.render(ocam, 6, traversal=0)
        assert st["exact_walk"] == want
        assert_same_frame(img, ref)
        assert st["rays"] == ost["rays"] and st["tri_tests"] > 0      # the lens does see the mesh
        _, wst = osc.use_product_walk(scene, fast=not want).render(ocam, 6, traversal=2)
        for k in ("interior_visits", "tri_tests", "plane_tests"):
            assert st[k] == wst[k], (want, k)
    # the local-pool route makes neither bet whatever is asked
    scene, cam, osc, ocam = both(scenes.cook_torrance_spheres_metallic, 48, 32, 2)
    _, st = rayrs_amd.render(scene, cam, 2, fast_traversal=True)
    assert st["local_pool"] == 1 and st["exact_walk"] == 1


def test_degenerate_and_extreme_primitives_render_the_oracles_frame():
    """Points, segments, flat and sliver triangles, a sphere of radius 1e-300, one of 1e30 and a triangle beyond f32's range
    (tests/test_bvh_builder.py degenerate_objects; f64 records): the default walk, the exact walk and the oracle's recursion
    render the same frame, with the oracle's counters for the tree each walks."""
    from test_bvh_builder import degenerate_objects
    from rayrs_amd.api import BvhHeuristic, Material, Emission
    objs = degenerate_objects()
    for o in objs[::3]:
        o.mat = Material.LambertianDiffuse((0.7, 0.6, 0.5))
    cam_args = ((3.0, 2.5, 5.0), (0.0, 1.0, 0.0), (0.0, 1.0, 0.0), 55.0, 56 / 254.0, 40 / 254.0, 100)
    scene = rayrs_amd.Scene(objs, 1e-6, 1e6, BvhHeuristic.Sah(1000), HDRI, device=0)
    assert not scene.info()["compact"]
    cam = rayrs_amd.Camera(*cam_args)
    osc = _oracle.OracleScene(objs, 1e-6, 1e6, BvhHeuristic.Sah(1000), HDRI)
    ocam = _oracle.OracleCamera(*cam_args)
    ref, ost = osc.render(ocam, 6, traversal=0)
    for exact in (False, True):
        img, st = rayrs_amd.render(scene, cam, 6, out_f64=True, count_work=True, exact_traversal=exact)
        assert st["exact_walk"] == int(exact) and st["rays"] == ost["rays"]
        assert_same_frame(img, ref)
        _, wst = osc.use_product_walk(scene, fast=not exact).render(ocam, 6, traversal=2)
        for k in ("interior_visits", "tri_tests", "sphere_tests", "plane_tests"):
            assert st[k] == wst[k], (exact, k)


def test_the_culling_walk_on_the_gate_tree_is_still_there_for_comparisons():
    """rayrs_lab.h gate_tree: rounds 2 and 3 walked the reference's groups behind their gating boxes with closest-hit
    culling; scripts/ubench/exact_cost.py prices the fast walk's own tree against it.  Same frame, the oracle's counters."""
    scene, cam, osc, ocam = both(lambda: scenes.mesh_scene(4), 96, 64, 4)
    a, sa = rayrs_amd.render(scene, cam, 4, out_f64=True, count_work=True, fast_traversal=True)
    scene.lab_set(gate_tree=1)
    b, sb = rayrs_amd.render(scene, cam, 4, out_f64=True, count_work=True, fast_traversal=True)
    scene.lab_set()
    assert_same_frame(a, b)
    _, wst = osc.use_walk_tree(scene, gate=True).render(ocam, 4, traversal=2)
    for k in ("rays", "interior_visits", "tri_tests", "plane_tests"):
        assert sb[k] == wst[k], k
    assert sa["tri_tests"] < sb["tri_tests"]  # what the leaf boxes are for


def test_render_multi_rejects_bad_handles():
    from rayrs_amd import _ffi
    scene, cam, osc, ocam = both(scenes.diffuse_single_sphere, 16, 16, 1)
    with pytest.raises(_ffi.RayrsError) as e:
        rayrs_amd.render_multi([scene, scene], cam, 1)   # the same handle twice: one render in flight per handle
    assert e.value.status == -1
    host_only = rayrs_amd.Scene(scenes.diffuse_single_sphere()[1], 1e-6, 1e6, scenes.SAH_1000, HDRI, device=-1)
    with pytest.raises(_ffi.RayrsError) as e:
        rayrs_amd.render_multi([scene, host_only], cam, 1)
    assert e.value.status == -4


def test_glass_on_a_triangle_mesh():
    """test_scenes.rs:165-168 (glass_suzanne; the asset is not in the repository): a dielectric on
    triangles -- constant geometric normals, entering/exiting decided by the sign of n.v alone, back
    faces never culled (geometry.rs:359-379), paths bouncing inside the closed mesh."""
    glass = lambda: scenes.mesh_scene(3, rayrs_amd.Material.Glass((0.8, 0.8, 0.8), 1.45))
    scene, cam, osc, ocam = both(glass, 64, 48, 8)
    img, st = rayrs_amd.render(scene, cam, 8, 50, out_f64=True)
    ref, ost = osc.render(ocam, 8, 50, traversal=0)
    assert st["rays"] == ost["rays"] and st["rays"] > 3 * 64 * 48 * 8 // 2  # refraction chains: many queries per path
    assert_same_frame(img, ref)
    frosted = lambda: scenes.mesh_scene(3, rayrs_amd.Material.CookTorranceGlass((1, 1, 1), 0.05, 1.45))
    scene, cam, osc, ocam = both(frosted, 48, 32, 4)
    img, st = rayrs_amd.render(scene, cam, 4, 50, out_f64=True)
    ref, ost = osc.render(ocam, 4, 50, traversal=0)
    assert st["rays"] == ost["rays"]
    assert_same_frame(img, ref)


def _mesh_from(origin, lookat):
    def fn():
        cam_args, objs, heur = scenes.mesh_scene(3, area_light=True)
        return (origin, cam_args[1], lookat) + tuple(cam_args[3:]), objs, heur
    return fn


def test_primary_rays_that_miss_the_root_box_are_answered_where_they_are_made():
    """bvh.rs:394: a ray that misses the root Node's box is a Miss at once.  The kernels that start a
    sample test that box themselves and add the background there (wavefront.hip next_sample) instead of
    sending the ray through the traversal and miss kernels.  Horizon through the frame: part of the
    primary rays take that way, part do not, within the same items; the frame, the ray count and the
    number of escaped paths must not change."""
    scene, cam, osc, ocam = both(_mesh_from((0.0, 6.0, 10.0), (0.0, 5.5, 0.0)), 64, 48, 12)
    box = osc.bbox()[0]
    missed = 0
    for i in range(0, 48, 4):
        for j in range(0, 64, 4):
            o, d = ocam.primary_ray(48 - i, 64 - j, 77 + i * 64 + j)[:2]
            missed += not _oracle.aabb_intersect(box, o, d, 1e-6, 1e6)
    assert 0.2 < missed / (12 * 16) < 0.8  # the case is what it says
    for chunk in (0, 4, 5):
        img, st = rayrs_amd.render(scene, cam, 12, 50, sample_chunk=chunk, out_f64=True, count_work=True)
        ref, ost = osc.use_product_walk(scene).render(ocam, 12, 50, sample_chunk=chunk, traversal=2)
        for k in ("rays", "paths", "escaped_paths", "interior_visits", "tri_tests", "plane_tests"):
            assert st[k] == ost[k], (chunk, k)
        assert 0.3 * 64 * 48 * 12 < st["direct_rays"] < 0.7 * 64 * 48 * 12
        assert_same_frame(img, ref)


def test_a_frame_of_sky_needs_no_path_rounds():
    """Camera turned away from the scene: every primary ray misses the root box, so every sample is
    finished by the kernel that starts it; the slots never hold a ray and the frame takes the first batch
    of (empty) rounds only, however many samples a pixel has."""
    scene, cam, osc, ocam = both(_mesh_from((0.0, 5.0, 10.0), (0.0, 9.0, 20.0)), 40, 24, 96)
    img, st = rayrs_amd.render(scene, cam, 96, 50, out_f64=True)  # one item per pixel: 96 samples in a row
    ref, ost = osc.render(ocam, 96, 50, traversal=0)
    assert st["rays"] == ost["rays"] == st["paths"] == 40 * 24 * 96 == st["escaped_paths"] == st["direct_rays"]
    assert st["kernel_launches"] <= 32  # two batches of 16 enqueued before the host sees the pool empty
    assert_same_frame(img, ref)


def _random_scene(seed):
    """Spheres, rectangles, triangles and boxes at coordinates that are not f32 values (the f64 record
    layout), every material kind, emitters among them, camera somewhere around."""
    from rayrs_amd.api import Axis, Emission, Fresnel, Material, Object
    r = np.random.default_rng(seed)
    col = lambda: tuple(float(x) for x in r.uniform(0.2, 1.0, 3))

    def material():
        k = int(r.integers(0, 10))
        a, ior = float(r.uniform(0.02, 0.6)), float(r.uniform(1.1, 1.9))
        return [lambda: Material.LambertianDiffuse(col()), lambda: Material.Reflect(col()),
                lambda: Material.Refract(col(), ior), lambda: Material.Glass(col(), ior),
                lambda: Material.CookTorrance(col(), a, Fresnel.SchlickMetallic(col())),
                lambda: Material.CookTorrance(col(), a, Fresnel.SchlickDielectric(ior)),
                lambda: Material.CookTorranceRefract(col(), a, ior), lambda: Material.CookTorranceGlass(col(), a, ior),
                lambda: Material.Plastic(col(), col(), a, ior), lambda: Material.NoReflect()][k]()

    def emission():
        return Emission.new(float(r.uniform(0.5, 4.0)), col()) if r.uniform() < 0.2 else Emission.Dark()

    objs = [Object.plane(Axis.Y, -8.0, 8.0, -8.0, 8.0, 0.0, material(), Emission.Dark())]
    for i in range(int(r.integers(6, 40))):
        c = r.uniform(-3.0, 3.0, 3)
        c[1] = abs(c[1]) + 0.1
        kind = int(r.integers(0, 4))
        if kind == 0:
            objs.append(Object.sphere(float(r.uniform(0.1, 0.9)), c, material(), emission()))
        elif kind == 1:
            objs.append(Object.plane(int(r.integers(0, 6)), c[0], c[0] + float(r.uniform(0.2, 2.0)), c[1],
                                     c[1] + float(r.uniform(0.2, 2.0)), c[2], material(), emission()))
        elif kind == 2:
            objs.append(Object.triangle(c, c + r.uniform(-1.5, 1.5, 3), c + r.uniform(-1.5, 1.5, 3), material(),
                                        emission()))
        else:
            objs += Object.box_geom(c, c + r.uniform(0.2, 1.2, 3), material(), emission())
    origin = r.uniform(-7.0, 7.0, 3)
    origin[1] = float(r.uniform(0.5, 6.0))
    lookat = r.uniform(-1.0, 1.0, 3)
    lookat[1] = float(r.uniform(0.0, 2.0))
    cam = (tuple(origin), (0.0, 1.0, 0.0), tuple(lookat), float(r.uniform(30.0, 100.0)), 3.84, 2.16, 100)
    return cam, objs, scenes.SAH_1000


@pytest.mark.parametrize("seed", [11, 12, 13, 14, 15, 16, 17, 18])
def test_random_scenes_render_the_oracles_frame(seed):
    """Nothing in these scenes was chosen by hand: primitives of all kinds behind all nine materials, in the
    f64 record layout, seen from anywhere (also from inside boxes and through the horizon)."""
    scene, cam, osc, ocam = both(lambda: _random_scene(seed), 40, 24, 6)
    assert not scene.info()["compact"]
    chunk = (0, 4)[seed & 1]
    img, st = rayrs_amd.render(scene, cam, 6, 50, seed=seed, sample_chunk=chunk, out_f64=True, count_work=True)
    ref, ost = osc.use_product_walk(scene).render(ocam, 6, 50, seed=seed, sample_chunk=chunk, traversal=2)
    for k in ("rays", "paths", "escaped_paths", "interior_visits", "tri_tests", "sphere_tests", "plane_tests",
              "nan_pixels", "neg_pixels"):
        assert st[k] == ost[k], k
    assert_same_frame(img, ref)
    ref0, ost0 = osc.render(ocam, 6, 50, seed=seed, sample_chunk=chunk, traversal=0)  # the reference's recursion
    assert ost0["rays"] == st["rays"]
    assert_same_frame(img, ref0)


def test_light_side_array_eager_and_on_demand():
    """A path's light is +0 until it meets an emitter, and lives in a side array only from then on (wavefront.h
    PathSlot).  Where a surface emits the kernels request the side entry with the slot (eager); elsewhere they
    would fetch it on demand.  rayrs_lab.h eager_light forces the eager kernels on a scene without emitters:
    same frame.  The scene WITH an emissive rectangle exercises the stored light, and must equal the oracle."""
    for fn, chunk in ((lambda: scenes.mesh_scene(3), 4), (scenes.material_test, 0), (scenes.glass_single_sphere, 5)):
        scene, cam, osc, ocam = both(fn, 72, 40, 12)
        demand, st1 = rayrs_amd.render(scene, cam, 12, 50, sample_chunk=chunk, out_f64=True)
        scene.lab_set(eager_light=1)
        eager, st2 = rayrs_amd.render(scene, cam, 12, 50, sample_chunk=chunk, out_f64=True)
        ref, ost = osc.render(ocam, 12, 50, sample_chunk=chunk, traversal=0)
        for k in ("rays", "paths", "escaped_paths", "nan_pixels", "neg_pixels"):
            assert st1[k] == st2[k] == ost[k], k
        assert_same_frame(demand, ref)
        assert_same_frame(eager, ref)
    # emitters of all kinds of material, light picked up mid-path and carried over many bounces
    from rayrs_amd.api import Axis, Emission, Material, Object
    def lit():
        cam_args, objs, heur = scenes.cook_torrance_spheres_metallic()
        objs = list(objs)
        objs.append(Object.plane(Axis.YRev, -4.0, 4.0, -4.0, 4.0, 5.0, Material.LambertianDiffuse((0.9, 0.9, 0.9)),
                                 Emission.new(3.0, (1.0, 0.8, 0.6))))
        objs.append(Object.sphere(0.6, (0.0, 3.0, 2.0), Material.Glass((1.0, 1.0, 1.0), 1.5), Emission.new(0.7, (0.2, 0.4, 1.0))))
        return cam_args, objs, heur
    scene, cam, osc, ocam = both(lit, 80, 40, 16)
    img, st = rayrs_amd.render(scene, cam, 16, 50, sample_chunk=4, out_f64=True)
    ref, ost = osc.render(ocam, 16, 50, sample_chunk=4, traversal=0)
    assert st["rays"] == ost["rays"]
    assert_same_frame(img, ref)


def test_two_different_frames_back_to_back_on_one_scene():
    """The path pool is kept between renders and wf_init_kernel rewrites the state bytes only: an IDLE slot's line still
    holds the previous frame's item, cursor and light bits.  Nothing may read them: a larger frame, then a
    smaller one with another spp, chunk and seed (fewer slots than before: most of the old lines stay stale), then the
    first again, each bit-identical to the oracle's frame and to its own first render."""
    cam_args, objs, heur = scenes.mesh_scene(3, area_light=True)   # an emitter: light bits get set
    scene = rayrs_amd.Scene(objs, 1e-6, 1e6, heur, HDRI, device=0)
    osc = _oracle.OracleScene(objs, 1e-6, 1e6, heur, HDRI)
    frames = []
    for (w, h, spp, chunk, seed) in ((96, 64, 8, 4, 1), (40, 24, 5, 0, 2), (96, 64, 8, 4, 1), (64, 40, 12, 4, 3)):
        ca = scenes.camera_for_resolution(cam_args, w, h)
        img, st = rayrs_amd.render(scene, rayrs_amd.Camera(*ca), spp, 50, seed=seed, sample_chunk=chunk, out_f64=True)
        ref, ost = osc.render(_oracle.OracleCamera(*ca), spp, 50, seed=seed, sample_chunk=chunk)
        assert st["rays"] == ost["rays"] and st["paths"] == ost["paths"]
        assert_same_frame(img, ref)
        frames.append(img)
    assert_same_frame(frames[0], frames[2])
    # the same with a pool far smaller than the items (slots are reused within a frame too)
    scene.set_tuning(pool_slots=2048)
    ca = scenes.camera_for_resolution(cam_args, 96, 64)
    img, _ = rayrs_amd.render(scene, rayrs_amd.Camera(*ca), 8, 50, seed=1, sample_chunk=4, out_f64=True)
    assert_same_frame(img, frames[0])
