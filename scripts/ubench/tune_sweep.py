"""Same-box sweep of rayrs_tuning settings on one scene built once (development aid).
usage: python scripts/ubench/tune_sweep.py <config> <res> <spp> "k=v,k=v" "k=v" ...   ("" = defaults)
Each setting is rendered twice in the order A B C ... C B A; prints trace / traversal ms per render.
TILE_RANKS=n in the environment renders rank 0's share of n (what one GPU of n does); CHUNK=n another sample chunk."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import rayrs_amd
from rayrs_amd import scenes, procedural

cfg, res, spp = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
settings = sys.argv[4:] or [""]
cam_args, objs, heur, _, mb = scenes.config(cfg)
cam_args = scenes.camera_for_resolution(cam_args, res, res)
scene = rayrs_amd.Scene(objs, 1e-6, 1e6, heur, procedural.make_hdri(1024, 512), device=0)
cam = rayrs_amd.Camera(*cam_args)
print("compact", scene.info()["compact"], "n_wide", scene.info()["n_wide"], flush=True)
chunk = int(os.environ.get("CHUNK", "0")) or rayrs_amd.frame_sample_chunk(res, res, spp)  # CHUNK=n: another sample chunk
rayrs_amd.render(scene, cam, 4, mb, sample_chunk=0)  # warm
ref = None
for s in settings + settings[::-1]:
    kw = {k: int(v) for k, v in (kv.split("=") for kv in s.split(",") if kv)}
    scene.set_tuning(**kw)
    img, st = rayrs_amd.render(scene, cam, spp, mb, sample_chunk=chunk, tile_ranks=int(os.environ.get("TILE_RANKS", "1")))
    if ref is None:
        ref = img.copy()
    same = bool((img.view("u4") == ref.view("u4")).all())
    print(f"[{s or 'defaults':40s}] trace {st['trace_ms']:8.1f} ms  trav {st['kernel_ms']:8.1f} ms  rounds {st['kernel_launches']:4d}  "
          f"Mray/s {st['rays'] / st['trace_ms'] / 1e3:8.1f}  hit {st['hit_ms']:7.1f} miss {st['miss_ms']:7.1f}  same_bits={same}", flush=True)
