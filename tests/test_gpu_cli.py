"""The C++ command line (rayrs_amd/rayrs, the counterpart of rayrs/src/main.rs) end to end on the GPU."""
import os
import subprocess

import numpy as np
import pytest

import rayrs_amd
from rayrs_amd import io, procedural, scenes

pytestmark = pytest.mark.gpu
CLI = os.path.join(os.path.dirname(os.path.abspath(rayrs_amd.__file__)), "rayrs")


def test_cli_writes_the_same_png_as_the_library(tmp_path):
    hdri = procedural.make_hdri(128, 64)
    io.save_hdr(tmp_path / "env.hdr", hdri)
    r = subprocess.run([CLI, str(tmp_path / "env.hdr"), "6", "--seed", "77"], cwd=tmp_path, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "Time taken material_test:" in r.stdout and "Clamped pixels:" in r.stdout  # main.rs:96-100, image.rs:218
    png = (tmp_path / "material_test.png").read_bytes()
    assert png[:8] == b"\x89PNG\r\n\x1a\n"
    out_hdr = io.load_hdr(tmp_path / "material_test.hdr")
    # the library, same scene (test_scenes.rs:276-331: 1221 x 159 px), same HDRI as decoded from the file
    cam_args, objs, heur = scenes.material_test()
    env = io.load_hdr(tmp_path / "env.hdr")
    scene = rayrs_amd.Scene(objs, 1e-6, 1e6, heur, env, device=0)
    cam = rayrs_amd.Camera(*cam_args)
    assert (cam.x_pixels(), cam.y_pixels()) == (1221, 159) == (out_hdr.shape[1], out_hdr.shape[0])
    img, st = rayrs_amd.render(scene, cam, 6, 50, seed=77)
    want, _ = io.to_raw_bytes(img)
    import zlib, struct
    pos, idat = 8, b""
    while pos < len(png):
        (n,) = struct.unpack(">I", png[pos:pos + 4])
        if png[pos + 4:pos + 8] == b"IDAT":
            idat += png[pos + 8:pos + 8 + n]
        pos += 12 + n
    rows = np.frombuffer(zlib.decompress(idat), dtype=np.uint8).reshape(159, 1 + 1221 * 3)
    assert np.array_equal(rows[:, 1:].reshape(159, 1221, 3), want)


def test_cli_usage_and_spp_fallback(tmp_path):
    r = subprocess.run([CLI], capture_output=True, text=True)
    assert r.returncode == 1 and "Usage: rayrs hdri_path [spp]" in r.stderr  # main.rs:126-128
    r = subprocess.run([CLI, str(tmp_path / "nope.hdr")], capture_output=True, text=True)
    assert r.returncode == 1


def test_cli_on_two_logical_gpus_writes_the_same_image(tmp_path):
    """--devices 0,0: rayrs_render_multi through the command line (--gpus N names devices 0..N-1)."""
    hdri = procedural.make_hdri(64, 32)
    io.save_hdr(tmp_path / "env.hdr", hdri)
    outs = []
    for extra in ([], ["--devices", "0,0"]):
        d = tmp_path / ("multi" if extra else "single")
        d.mkdir()
        r = subprocess.run([CLI, str(tmp_path / "env.hdr"), "4", "--scene", "diffuse_single_sphere", "--seed", "5"]
                           + extra, cwd=d, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        outs.append((d / "diffuse_single_sphere.png").read_bytes())
    assert outs[0] == outs[1]
