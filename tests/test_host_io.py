"""File formats either side of the path: PLY/OBJ ingest,
Radiance .hdr, the reference's 8-bit conversion, PPM/PNG writers.  CPU only."""
import os
import struct
import zlib

import numpy as np
import pytest

from rayrs_amd import _ffi, io, procedural


def test_ply_binary_and_ascii_round_trip_exactly(tmp_path):
    verts, idx = procedural.blob_mesh(2)
    for binary in (True, False):
        p = tmp_path / f"m{int(binary)}.ply"
        io.save_ply(p, verts, idx, binary=binary)
        v2, i2 = io.load_ply(p)
        assert v2.dtype == np.float32 and np.array_equal(v2.view(np.uint32), verts.view(np.uint32))
        assert np.array_equal(i2, idx)


def test_ply_quads_extra_properties_and_double_coordinates(tmp_path):
    p = tmp_path / "quad.ply"
    p.write_text("ply\nformat ascii 1.0\ncomment made by hand\nelement vertex 4\nproperty double x\nproperty double y\n"
                 "property double z\nproperty uchar red\nelement face 1\nproperty list uchar uint vertex_index\n"
                 "end_header\n0 0 0 255\n1 0 0 255\n1 1 0 0\n0 1 0 0\n4 0 1 2 3\n")
    v, i = io.load_ply(p)
    assert v.shape == (4, 3) and i.tolist() == [[0, 1, 2], [0, 2, 3]]  # fan, winding kept


def test_ply_errors(tmp_path):
    with pytest.raises(_ffi.RayrsError):
        io.load_ply(tmp_path / "missing.ply")
    bad = tmp_path / "bad.ply"
    bad.write_text("ply\nformat ascii 1.0\nelement vertex 1\nproperty float x\nproperty float y\nproperty float z\n"
                   "element face 1\nproperty list uchar int vertex_indices\nend_header\n0 0 0\n3 0 1 2\n")
    with pytest.raises(_ffi.RayrsError, match="out of range"):
        io.load_ply(bad)


def test_obj_loader_follows_the_reference(tmp_path):
    p = tmp_path / "t.obj"
    p.write_text("# comment\nv 0 0 0\nv 1 0 0\nv 0 1 0\nvn 0 0 1\nv 0.5 0.25 1e-3\nf 1 2 3\nf 1 3 4\n")
    v, i = io.load_obj(p)
    assert v.dtype == np.float64 and v.shape == (4, 3) and v[3].tolist() == [0.5, 0.25, 1e-3]
    assert i.tolist() == [[0, 1, 2], [0, 2, 3]]  # 1-based -> 0-based (wavefront_obj.rs:37-41)


def test_obj_sphere_loader_follows_the_reference(tmp_path):
    """wavefront_obj::load_obj_file_spheres (wavefront_obj.rs:46-64): one sphere of the given radius per `v` line; `f` lines --
    even ones load_obj_file would choke on -- and everything else are skipped."""
    p = tmp_path / "s.obj"
    p.write_text("# comment\nv 0 0 0\nv 1 0.5 -2\nvn 0 0 1\nf 1 2 99\nf nonsense\nv 0.5 0.25 1e-3\n")
    c = io.load_obj_spheres(p, 0.1)
    assert c.dtype == np.float64 and c.tolist() == [[0, 0, 0], [1, 0.5, -2], [0.5, 0.25, 1e-3]]
    from rayrs_amd.api import Emission, Material
    objs = io.load_obj_spheres(p, 0.1, Material.NoReflect(), Emission.Dark())
    assert [o.kind for o in objs] == ["sphere"] * 3 and objs[1].radius == 0.1 and objs[1].origin == (1.0, 0.5, -2.0)
    bad = tmp_path / "bad.obj"
    bad.write_text("v 1 2\n")
    with pytest.raises(_ffi.RayrsError):
        io.load_obj_spheres(bad, 1.0)
    with pytest.raises(_ffi.RayrsError):
        io.load_obj_spheres(tmp_path / "missing.obj", 1.0)


def test_hdr_round_trip_within_rgbe_precision(tmp_path):
    img = procedural.make_hdri(64, 32)
    img[0, 0] = 0.0
    p = tmp_path / "e.hdr"
    io.save_hdr(p, img)
    back = io.load_hdr(p)
    assert back.shape == img.shape
    m = img.max(axis=2, keepdims=True)
    assert np.all(np.abs(back - img) <= m / 128.0 + 1e-7)  # 8-bit mantissa relative to the brightest channel
    assert np.all(back[0, 0] == 0.0)


def test_hdr_reads_new_style_rle(tmp_path):
    w, h = 16, 2
    body = b""
    for y in range(h):
        body += bytes([2, 2, 0, w])
        for c, val in enumerate((128, 64, 32, 129)):  # r, g, b, e planes; exponent 129 -> scale 2^(129-136)
            body += bytes([128 + w, val])             # one run of w
    p = tmp_path / "rle.hdr"
    p.write_bytes(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 2 +X 16\n" + body)
    img = io.load_hdr(p)
    assert img.shape == (2, 16, 3) and np.allclose(img[0, 0], [1.0, 0.5, 0.25])


def test_to_raw_bytes_matches_image_rs():
    rgb = np.array([[[0.5, 0.0, 1.0], [2.0, -0.5, float("nan")], [0.25, 1.0, 0.999]]], dtype=np.float32)
    out, counts = io.to_raw_bytes(rgb, 1.0 / 2.2)
    # vecmath.rs:363-366: 0.5.powf(1/2.2) == 0.7297400528407231 -> (255.99 * .) as u8 == 186
    assert out[0, 0].tolist() == [int(255.99 * 0.7297400528407231), 0, 255]
    assert out[0, 1].tolist() == [255, 0, 255]   # clip; NaN.min(1).max(0) == 1.0 in Rust, `as u8` saturates
    assert counts == {"clamped": 1, "nan": 1, "negative": 1}


def test_ppm_and_png_are_valid(tmp_path):
    rgb = procedural.make_hdri(40, 24)
    b, _ = io.to_raw_bytes(rgb)
    io.save_ppm(tmp_path / "a.ppm", b)
    data = (tmp_path / "a.ppm").read_bytes()
    assert data.startswith(b"P6\n40 24\n255\n") and data[len(b"P6\n40 24\n255\n"):] == b.tobytes()
    io.save_png(tmp_path / "a.png", b)
    png = (tmp_path / "a.png").read_bytes()
    assert png[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, seen = 8, b"", []
    while pos < len(png):
        (n,) = struct.unpack(">I", png[pos:pos + 4])
        typ, payload = png[pos + 4:pos + 8], png[pos + 8:pos + 8 + n]
        (crc,) = struct.unpack(">I", png[pos + 8 + n:pos + 12 + n])
        assert zlib.crc32(typ + payload) == crc
        seen.append(typ)
        if typ == b"IHDR":
            assert struct.unpack(">IIBBBBB", payload) == (40, 24, 8, 2, 0, 0, 0)
        if typ == b"IDAT":
            idat += payload
        pos += 12 + n
    assert seen == [b"IHDR", b"IDAT", b"IEND"]
    raw = zlib.decompress(idat)
    rows = np.frombuffer(raw, dtype=np.uint8).reshape(24, 1 + 40 * 3)
    assert np.all(rows[:, 0] == 0) and np.array_equal(rows[:, 1:].reshape(24, 40, 3), b)


def test_mesh_scene_through_ply_is_the_same_scene(tmp_path):
    """configs[2]/[4] take their mesh through a PLY file: same objects, same tree."""
    import rayrs_amd
    from rayrs_amd import scenes
    hdri = procedural.make_hdri(32, 16)
    _, direct, heur = scenes.mesh_scene(2)
    _, via_ply, _ = scenes.mesh_scene(2, ply_path=tmp_path / "mesh.ply")
    a = rayrs_amd.Scene(direct, 1e-6, 1e6, heur, hdri, device=-1).export_bvh()
    b = rayrs_amd.Scene(via_ply, 1e-6, 1e6, heur, hdri, device=-1).export_bvh()
    assert all(np.array_equal(x, y) for x, y in zip(a, b))


def test_ply_header_is_not_trusted(tmp_path):
    """a PLY header claiming 2^32-1 vertices, a face list longer than the file, an index that is
    negative, fractional or huge, a double coordinate an f32 cannot hold -- all refused with RAYRS_IO_ERROR
    (-6), none crashes or allocates what the header asks for."""
    from rayrs_amd import _ffi

    def status(text, binary_tail=b""):
        p = tmp_path / "bad.ply"
        p.write_bytes(text.encode() + binary_tail)
        try:
            io.load_ply(p)
        except _ffi.RayrsError as e:
            return e.status
        return 0

    head = "ply\nformat ascii 1.0\nelement vertex {nv}\nproperty float x\nproperty float y\nproperty float z\n" \
           "element face {nf}\nproperty list uchar int vertex_indices\nend_header\n"
    good = head.format(nv=3, nf=1) + "0 0 0\n1 0 0\n0 1 0\n3 0 1 2\n"
    assert status(good) == 0
    assert status(head.format(nv=4294967295, nf=1) + "0 0 0\n") == -6            # count the body cannot hold
    assert status(head.format(nv=3, nf=1) + "0 0 0\n1 0 0\n0 1 0\n3 0 1 -2\n") == -6   # negative index
    assert status(head.format(nv=3, nf=1) + "0 0 0\n1 0 0\n0 1 0\n3 0 1 1.5\n") == -6  # fractional index
    assert status(head.format(nv=3, nf=1) + "0 0 0\n1 0 0\n0 1 0\n3 0 1 7\n") == -6    # beyond the vertices
    dbl = head.replace("property float x", "property double x")
    assert status(dbl.format(nv=3, nf=1) + "0.1 0 0\n1 0 0\n0 1 0\n3 0 1 2\n") == -6   # 0.1 is not an f32 value
    assert status(dbl.format(nv=3, nf=1) + "0.5 0 0\n1 0 0\n0 1 0\n3 0 1 2\n") == 0    # 0.5 is
    binh = head.replace("ascii", "binary_little_endian")
    assert status(binh.format(nv=1000000, nf=0), b"\0" * 36) == -6                     # truncated binary body


def test_writers_report_failure(tmp_path):
    from rayrs_amd import _ffi
    img = np.zeros((4, 4, 3), dtype=np.float32)
    with pytest.raises(_ffi.RayrsError):
        io.save_hdr(tmp_path / "no_such_dir" / "x.hdr", img)
    if os.path.exists("/dev/full"):  # every write fails with ENOSPC
        with pytest.raises(_ffi.RayrsError):
            io.save_hdr("/dev/full", np.zeros((64, 64, 3), dtype=np.float32))
