"""Synthetic inputs: the reference ships no HDRI and no mesh (its .gitignore
excludes *.hdr, *.ply, *.obj), so the environment map and the meshes of the
benchmark configurations are generated here.

Only +, -, *, /, sqrt on float64 are used (then one cast to float32), so the
same bytes come out on every machine: no libm call is involved.
"""
import numpy as np


def make_hdri(width: int = 1024, height: int = 512) -> np.ndarray:
    """Smooth sky gradient + a bright sun lobe + a warm ground; values in [0, 3]
    (the reference clips its HDRI to that range, rayrs/src/main.rs:43)."""
    v = (np.arange(height, dtype=np.float64) + 0.5) / height          # 0 top .. 1 bottom
    u = (np.arange(width, dtype=np.float64) + 0.5) / width
    V, U = np.meshgrid(v, u, indexing="ij")
    up = 1.0 - 2.0 * V                                                # +1 zenith, -1 nadir
    sky = np.clip(up, 0.0, 1.0)
    ground = np.clip(-up, 0.0, 1.0)
    horizon = 1.0 - np.abs(up)
    horizon = horizon * horizon * horizon
    r = 0.25 + 0.20 * horizon + 0.10 * ground
    g = 0.40 + 0.25 * horizon + 0.08 * ground - 0.10 * sky
    b = 0.75 - 0.25 * ground + 0.20 * horizon + 0.15 * sky
    # sun lobe: (1 - d^2/r^2)^4 inside radius r in (u, v) space, u wraps
    du = U - 0.30
    du = du - np.round(du)
    dv = V - 0.22
    d2 = (du * du) * 4.0 + dv * dv
    lobe = np.clip(1.0 - d2 / (0.06 * 0.06), 0.0, 1.0)
    lobe = lobe * lobe
    lobe = lobe * lobe
    r = r + 2.6 * lobe
    g = g + 2.4 * lobe
    b = b + 2.0 * lobe
    # a fainter, wider fill light on the other side
    du2 = U - 0.78
    du2 = du2 - np.round(du2)
    dv2 = V - 0.35
    d22 = (du2 * du2) * 4.0 + dv2 * dv2
    fill = np.clip(1.0 - d22 / (0.2 * 0.2), 0.0, 1.0)
    fill = fill * fill
    r = r + 0.5 * fill
    g = g + 0.45 * fill
    b = b + 0.4 * fill
    img = np.stack([r, g, b], axis=-1)
    return np.clip(img, 0.0, 3.0).astype(np.float32)


def _icosahedron():
    t = (1.0 + np.sqrt(5.0)) / 2.0
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                  [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], dtype=np.float64)
    v = v / np.sqrt((v * v).sum(axis=1, keepdims=True))
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2],
                  [10, 7, 6], [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5],
                  [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]], dtype=np.int64)
    return v, f


def icosphere(level: int):
    """Unit icosphere: 20 * 4**level triangles (counter-clockwise seen from outside)."""
    v, f = _icosahedron()
    for _ in range(level):
        a, b, c = f[:, 0], f[:, 1], f[:, 2]
        edges = np.concatenate([np.stack([a, b], 1), np.stack([b, c], 1), np.stack([c, a], 1)], axis=0)
        edges.sort(axis=1)
        key = edges[:, 0] * (len(v) + 1) + edges[:, 1]
        uniq, inv = np.unique(key, return_inverse=True)
        e0 = uniq // (len(v) + 1)
        e1 = uniq % (len(v) + 1)
        mid = v[e0] + v[e1]
        mid = mid / np.sqrt((mid * mid).sum(axis=1, keepdims=True))
        n0 = len(v)
        v = np.concatenate([v, mid], axis=0)
        m = inv.reshape(3, -1) + n0
        ab, bc, ca = m[0], m[1], m[2]
        f = np.concatenate([np.stack([a, ab, ca], 1), np.stack([b, bc, ab], 1), np.stack([c, ca, bc], 1),
                            np.stack([ab, bc, ca], 1)], axis=0)
    return v, f


def _cheb(n, x):
    t0, t1 = np.ones_like(x), x
    for _ in range(n - 1):
        t0, t1 = t1, 2.0 * x * t1 - t0
    return t1 if n > 0 else t0


def blob_mesh(level: int, center=(0.0, 1.25, 0.0), radius: float = 1.0):
    """Closed bumpy mesh standing in for the scanned models the configs name
    (no asset exists): an icosphere displaced radially by low- and
    high-frequency polynomial terms.  Returns (verts f32 (n,3), idx u32 (m,3)) --
    f32 because that is what a PLY file stores."""
    v, f = icosphere(level)
    x, y, z = v[:, 0], v[:, 1], v[:, 2]
    disp = (1.0 + 0.18 * (4.0 * x * y * z) + 0.10 * (x * x - z * z) * (3.0 * y)
            + 0.035 * _cheb(7, x) * _cheb(5, y) + 0.03 * _cheb(6, z) * _cheb(9, x)
            + 0.012 * _cheb(17, y) * _cheb(13, z))
    p = v * (radius * disp)[:, None] + np.asarray(center, dtype=np.float64)[None, :]
    return p.astype(np.float32), f.astype(np.uint32)
