"""Same-box sweep of rayrs_tuning / rayrs_lab.h settings on one scene built once (development aid).
usage: python scripts/ubench/tune_sweep.py <config> <res> <spp> "k=v,k=v" "k=v" ...   ("" = defaults)
Each setting is rendered twice in the order A B C ... C B A; prints trace / traversal ms per render.
TILE_RANKS=n in the environment renders one rank's share of n (what one GPU of n does; TILE_RANK=r which, default 0,
"all" = every rank in turn, slowest reported last); CHUNK=n another sample chunk; FAST=1 renders with
rayrs_render_params.fast_traversal; CAM=close the camera the mesh fills (scenes.MESH_CLOSE_CAM)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import rayrs_amd
from rayrs_amd import scenes, procedural

cfg, res, spp = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
settings = sys.argv[4:] or [""]
cam_args, objs, heur, _, mb = scenes.config(cfg)
if os.environ.get("CAM") == "close":  # the camera the mesh fills (bench.py's secondary line)
    cam_args = scenes.MESH_CLOSE_CAM
cam_args = scenes.camera_for_resolution(cam_args, res, res)
scene = rayrs_amd.Scene(objs, 1e-6, 1e6, heur, procedural.make_hdri(1024, 512), device=0)
cam = rayrs_amd.Camera(*cam_args)
print("compact", scene.info()["compact"], "n_wide", scene.info()["n_wide"], flush=True)
chunk = int(os.environ.get("CHUNK", "0")) or rayrs_amd.frame_sample_chunk(res, res, spp)  # CHUNK=n: another sample chunk
rayrs_amd.render(scene, cam, 4, mb, sample_chunk=0)  # warm
ref = None
for s in settings + settings[::-1]:
    kw = {k: int(v, 0) for k, v in (kv.split("=") for kv in s.split(",") if kv)}
    public = {k: v for k, v in kw.items() if k in ("pool_slots", "local_pool")}
    scene.set_tuning(**public)
    scene.lab_set(**{k: v for k, v in kw.items() if k not in public})
    n_ranks = int(os.environ.get("TILE_RANKS", "1"))
    which = os.environ.get("TILE_RANK", "0")
    ranks = range(n_ranks) if which == "all" else [int(which)]
    img, worst = None, None
    for r in ranks:
        img, st = rayrs_amd.render(scene, cam, spp, mb, sample_chunk=chunk, tile_rank=r, tile_ranks=n_ranks, out=img,
                                   fast_traversal=bool(int(os.environ.get("FAST", "0"))))
        if len(ranks) > 1:
            print(f"   rank {r}: trace {st['trace_ms']:8.1f} ms  trav {st['kernel_ms']:8.1f}  hit {st['hit_ms']:7.1f} miss {st['miss_ms']:7.1f} "
                  f"rounds {st['kernel_launches']:4d}  Mray/s {st['rays'] / st['trace_ms'] / 1e3:8.1f}", flush=True)
        if worst is None or st["trace_ms"] > worst["trace_ms"]:
            worst = st
    st = worst
    if ref is None:
        ref = img.copy()
    same = bool((img.view("u4") == ref.view("u4")).all())
    print(f"[{s or 'defaults':40s}] trace {st['trace_ms']:8.1f} ms  trav {st['kernel_ms']:8.1f} ms  rounds {st['kernel_launches']:4d}  "
          f"Mray/s {st['rays'] / st['trace_ms'] / 1e3:8.1f}  hit {st['hit_ms']:7.1f} miss {st['miss_ms']:7.1f}  same_bits={same}  "
          f"sha {__import__('hashlib').sha256(img.tobytes()).hexdigest()[:12]}", flush=True)
