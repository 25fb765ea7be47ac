"""Per-round kernel times of a frame (or of one rank's tile share) from the library's own HIP events
(rayrs_lab.h rayrs_lab_round_ms).  usage: python scripts/ubench/round_times.py <config> <res> <spp> [tile_ranks [tile_rank]]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import rayrs_amd
from rayrs_amd import scenes, procedural, _ffi
cfg, res, spp = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
ranks = int(sys.argv[4]) if len(sys.argv) > 4 else 1
rank = int(sys.argv[5]) if len(sys.argv) > 5 else 0
cam_args, objs, heur, _, mb = scenes.config(cfg)
cam_args = scenes.camera_for_resolution(cam_args, res, res)
scene = rayrs_amd.Scene(objs, 1e-6, 1e6, heur, procedural.make_hdri(1024, 512), device=0)
cam = rayrs_amd.Camera(*cam_args)
chunk = rayrs_amd.frame_sample_chunk(res, res, spp)
lab = {k: int(v, 0) for k, v in (kv.split("=") for kv in os.environ.get("LAB", "").split(",") if kv)}
if lab:
    scene.lab_set(**lab)
rayrs_amd.render(scene, cam, 4, mb)
img, st = rayrs_amd.render(scene, cam, spp, mb, sample_chunk=chunk, tile_rank=rank, tile_ranks=ranks)
L = _ffi.lib()
L.rayrs_lab_round_ms.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
n = L.rayrs_lab_round_ms(scene._h, None, 0)
t = np.zeros((n, 3), dtype=np.float32)
L.rayrs_lab_round_ms(scene._h, t.ctypes.data, n)
print(f"rank {rank} of {ranks}: trace {st['trace_ms']:.1f} ms, {n} rounds: trav {t[:,0].sum():.1f} hit {t[:,1].sum():.1f} miss {t[:,2].sum():.1f}")
for r in range(n):
    print(f"round {r:3d}  trav {t[r,0]:7.3f}  hit {t[r,1]:7.3f}  miss {t[r,2]:7.3f}")
