"""Development aid (CPU only): prices scheduling policies of the traversal kernel's waves on the rays a real frame makes.
The headline scene is built host-only (device=-1), its default-walk tree exported; the oracle path-traces the samples of a few
8x8 tiles (item order: pixel, chunk of 4 samples) and dumps every ray (orc_set_ray_dump); walk_sched_sim.c then plays the pool's
rounds and the waves' phases.  usage: python scripts/sim/walk_sched_sim.py [level=8] [tiles=4] [spp=1024]"""
import ctypes as C
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _oracle  # noqa: E402
import rayrs_amd  # noqa: E402
from rayrs_amd import procedural, scenes  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


class Params(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("policy", "refill_min", "leaf_min", "blocked_max", "queue_cap", "windows_per_wave",
                                          "n_slots", "int_min")]


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("rounds", "rays", "walk_rays", "int_steps", "int_lanes", "leaf_steps", "leaf_lanes",
                                          "leaf_prim_steps", "refills", "refill_lanes", "records", "prims", "max_queue",
                                          "queue_full_waits", "leaf_groups")]


def sim_lib():
    so = os.path.join(HERE, "walk_sched_sim.so")
    src = os.path.join(HERE, "walk_sched_sim.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-ffp-contract=off", "-o", so, src])
    L = C.CDLL(so)
    L.sim_run.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_double, C.c_double, C.c_void_p, C.c_uint64,
                          C.POINTER(Params), C.POINTER(Stats)]
    return L


def dump_rays(level, tiles, spp, res, cache):
    if os.path.exists(cache):
        z = np.load(cache)
        return z["rays"]
    if level == 8:
        cam_args, objs, heur, _, mb = scenes.config(5)
    else:
        cam_args, objs, heur = scenes.mesh_scene(level)
        mb = 50
    cam_args = scenes.camera_for_resolution(cam_args, res, res)
    hdri = procedural.make_hdri(1024, 512)
    t = time.time()
    osc = _oracle.OracleScene(objs, 1e-6, 1e6, heur, hdri)
    print(f"oracle scene built in {time.time() - t:.1f} s", flush=True)
    ocam = _oracle.OracleCamera(*cam_args)
    L = _oracle.lib()
    L.orc_set_ray_dump.argtypes = [C.c_void_p, C.c_uint64]
    L.orc_ray_dump_count.restype = C.c_uint64
    cap = tiles * 64 * spp * 12
    buf = np.zeros((cap, 7))
    L.orc_set_ray_dump(buf.ctypes.data, cap)
    rgb = (C.c_double * 3)()
    dummy_o = np.zeros(4, dtype=np.int64); dummy_t = np.zeros(4); dummy_thr = np.zeros((4, 3)); dummy_d = np.zeros(4, dtype=np.uint32)
    # tiles spread over the image: 8x8 pixels each, items in (pixel, chunk) order = all samples of a pixel back to back
    rng = np.random.default_rng(5)
    t = time.time()
    for ti in range(tiles):
        ty, tx = int(rng.integers(0, res // 8)), int(rng.integers(0, res // 8))
        for p in range(64):
            row, col = ty * 8 + p // 8, tx * 8 + p % 8
            for s in range(spp):
                L.orc_path_trace(osc._h, C.byref(ocam.desc), row, col, s, 0x5EED, mb, 0, 4, dummy_o.ctypes.data,
                                 dummy_t.ctypes.data, dummy_thr.ctypes.data, dummy_d.ctypes.data, rgb)
        print(f"tile {ti} ({ty},{tx}) done, {L.orc_ray_dump_count()} rays, {time.time() - t:.0f} s", flush=True)
    n = L.orc_ray_dump_count()
    L.orc_set_ray_dump(None, 0)
    rays = buf[:n].copy()
    np.savez(cache, rays=rays)
    return rays


def main():
    level = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    tiles = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    spp = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
    res = 2048
    global N_SLOTS
    N_SLOTS = int(sys.argv[4]) if len(sys.argv) > 4 else 32768
    cache = f"/tmp/walk_sim_rays_l{level}_t{tiles}_s{spp}.npz"
    rays = dump_rays(level, tiles, spp, res, cache)
    print(f"{len(rays)} rays, {int((rays[:, 6] == 0).sum())} paths", flush=True)
    if level == 8:
        cam_args, objs, heur, _, mb = scenes.config(5)
    else:
        cam_args, objs, heur = scenes.mesh_scene(level)
    prod = rayrs_amd.Scene(objs, 1e-6, 1e6, heur, procedural.make_hdri(64, 32), device=-1)
    info = prod.info()
    box, ref = prod.export_hot_tree()
    box, ref = np.ascontiguousarray(box, dtype=np.float64), np.ascontiguousarray(ref, dtype=np.uint32)
    root_rec = info["hot_root_ref"] & 0x3fffffff
    rb = np.array(info["root_box"], dtype=np.float64)
    print("hot tree records", info["hot_n_wide"], "root_rec", root_rec, "root box", rb, flush=True)
    L = sim_lib()

    def run(**kw):
        p = Params(policy=0, refill_min=56, leaf_min=32, blocked_max=64, queue_cap=8, windows_per_wave=43, n_slots=N_SLOTS, int_min=0)
        for k, v in kw.items():
            setattr(p, k, v)
        st = Stats()
        L.sim_run(box.ctypes.data, ref.ctypes.data, root_rec, rb.ctypes.data, 1e-6, 1e6, rays.ctypes.data, len(rays), C.byref(p), C.byref(st))
        wr = max(st.walk_rays, 1)
        ui = st.int_lanes / max(st.int_steps, 1) / 64
        ul = st.leaf_lanes / max(st.leaf_steps, 1) / 64
        # cost model from the counting build (profiles/r06_hot_group.txt (2)): interior step 8.65 k ticks, leaf step 17.6 k at ~4 primitives
        cost = st.int_steps * 8.65 + st.leaf_steps * 1.6 + st.leaf_prim_steps * 4.0
        cost_d = st.int_steps * 8.65 * 1.08 + st.leaf_steps * 2.2 + st.leaf_prim_steps * 4.0
        print(f"{kw}: walk rays {st.walk_rays / max(st.rays, 1):.3f} of rays, rec/walk ray {(st.records - st.rays) / wr:.2f} prims {st.prims / wr:.2f}  "
              f"int steps/ray {st.int_steps / wr:.4f} util {ui:.3f}  leaf steps/ray {st.leaf_steps / wr:.4f} util {ul:.3f} maxc {st.leaf_prim_steps / max(st.leaf_steps, 1):.2f}  "
              f"refills/ray {st.refills / wr:.4f} ({st.refill_lanes / max(st.refills, 1):.1f} lanes)  maxq {st.max_queue} fullwaits/ray {st.queue_full_waits / wr:.3f}  "
              f"cost/ray {cost / wr:.3f} (with overheads {cost_d / wr:.3f})", flush=True)
        return st

    run(policy=0)
    run(policy=0, refill_min=52)
    for cap in (8, 12):
        for lm in (32, 40, 48, 56):
            for bm in (8, 16, 24, 64):
                run(policy=1, queue_cap=cap, leaf_min=lm, blocked_max=bm)


if __name__ == "__main__":
    main()
