"""BASELINE.json's configurations at their FULL sizes on the GPU.  The oracle cannot
render a whole frame at these sizes in test time, so each full GPU frame is checked
(1) bit for bit against the oracle on a band of image rows rendered at the full
resolution and full spp (pixels are independent, so a band is an exact sub-problem),
and (2) through size-independent properties: the ray count equals the oracle's on
the band, no NaN/negative pixels, tile-sharded renders sum to the frame exactly."""
import numpy as np
import pytest

import _oracle
import rayrs_amd
from rayrs_amd import procedural, scenes

pytestmark = pytest.mark.gpu
HDRI = procedural.make_hdri(1024, 512)


def run_config(n, band_rows, tmp_path=None, r0=None):
    """Config n exactly as `python bench.py --config n` renders it: full resolution, full spp, and the
    sample chunk of rayrs_frame_sample_chunk (the rule bench.py, the CLI and this test share)."""
    cam_args, objs, heur, spp, mb = scenes.config(n, ply_path=(tmp_path / "mesh.ply") if tmp_path else None)
    scene = rayrs_amd.Scene(objs, 1e-6, 1e6, heur, HDRI, device=0)
    cam = rayrs_amd.Camera(*cam_args)
    H, W = cam.y_pixels(), cam.x_pixels()
    chunk = rayrs_amd.frame_sample_chunk(W, H, spp)
    assert chunk == {1: 4, 2: 4, 3: 4, 4: 16, 5: 4}[n]
    img, st = rayrs_amd.render(scene, cam, spp, mb, seed=0x5EED, sample_chunk=chunk, out_f64=True)
    assert st["paths"] == H * W * spp and st["nan_pixels"] == 0 and st["neg_pixels"] == 0
    osc = _oracle.OracleScene(objs, 1e-6, 1e6, heur, HDRI, builder=1)
    ocam = _oracle.OracleCamera(*cam_args)
    if r0 is None:
        r0 = H // 2 - band_rows // 2
    ref, ost = osc.render(ocam, spp, mb, seed=0x5EED, sample_chunk=chunk, rows=(r0, r0 + band_rows), traversal=0)
    band = img[r0:r0 + band_rows]
    want = ref[r0:r0 + band_rows]
    assert np.array_equal(band.view(np.uint64), want.view(np.uint64)), \
        f"{int((band != want).any(axis=2).sum())} band pixels differ"
    assert ost["paths"] == band_rows * W * spp
    return scene, cam, img, st, (spp, mb)


def test_config2_metallic_spheres_1024x1024_256spp():
    scene, cam, img, st, (spp, mb) = run_config(2, band_rows=8)
    # tile sharding at full size: three ranks, exact sum
    full32 = img.astype(np.float32)
    parts = [rayrs_amd.render(scene, cam, spp, mb, seed=0x5EED, sample_chunk=4, tile_rank=r, tile_ranks=3)[0]
             for r in range(3)]
    assert np.array_equal(parts[0] + parts[1] + parts[2], full32)


def test_config3_70k_triangle_ply_mesh_area_light_1024x1024_512spp(tmp_path):
    scene, cam, img, st, _ = run_config(3, band_rows=4, tmp_path=tmp_path)
    assert scene.info()["n_prims"] == 81922 and scene.info()["compact"] == 1
    assert img.max() > 1.0  # the emitter is visible


def test_config4_frosted_glass_depth32_2048x2048_4096spp():
    """configs[3] in full: 17.2 G paths, 28 G rays, depth limit 32, sample chunk 16 (the frame rule's answer).
    The oracle renders two full-spp rows through the spheres."""
    scene, cam, img, st, (spp, mb) = run_config(4, band_rows=2, r0=1024 + 96)
    assert (spp, mb) == (4096, 32)
    assert st["rays"] > 25_000_000_000


def test_config5_1m_triangle_ply_mesh_2048x2048_1024spp(tmp_path):
    scene, cam, img, st, _ = run_config(5, band_rows=2, tmp_path=tmp_path)
    assert scene.info()["n_prims"] == 1310721
    assert st["rays"] > 7_000_000_000
