"""layout.h TileOrder (the order in which a rank's tiles become items: vertical stripes, each walked row by row) is a
permutation of the rank's tiles for every frame shape, rank count and stripe width -- checked exhaustively on the host
for frames up to 40 x 12 tiles, ranks 1..9 and eight stripe widths (tests/tile_order_check.cpp; the GPU side renders with
it in tests/test_gpu_render.py::test_tile_order_is_a_permutation_of_the_ranks_tiles)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_tile_order_is_a_bijection(tmp_path):
    exe = tmp_path / "tile_order_check"
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", str(exe), os.path.join(ROOT, "tests", "tile_order_check.cpp")],
                   check=True, capture_output=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "bad 0" in r.stdout
