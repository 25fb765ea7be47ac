"""Where the hit and miss kernels' waves spend their time (development build: make -C rayrs_amd/csrc LAB=1).
usage: python scripts/ubench/shade_ticks.py <config> <res> <spp>"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import rayrs_amd
from rayrs_amd import scenes, procedural, _ffi
cfg, res, spp = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
cam_args, objs, heur, _, mb = scenes.config(cfg)
cam_args = scenes.camera_for_resolution(cam_args, res, res)
scene = rayrs_amd.Scene(objs, 1e-6, 1e6, heur, procedural.make_hdri(1024, 512), device=0)
cam = rayrs_amd.Camera(*cam_args)
chunk = rayrs_amd.frame_sample_chunk(res, res, spp)
img, st = rayrs_amd.render(scene, cam, spp, mb, sample_chunk=chunk)
L = _ffi.lib()
out = (C.c_uint64 * 16)()
L.rayrs_lab_ticks.argtypes = [C.c_void_p, C.c_void_p]
assert L.rayrs_lab_ticks(scene._h, out) == 0
t = list(out)
print(f"trace {st['trace_ms']:.1f} ms trav {st['kernel_ms']:.1f} hit {st['hit_ms']:.1f} miss {st['miss_ms']:.1f}")
hb = max(t[5], 1)
names = ["feed + issue next batch's slot loads", "wait for this batch's data + Material::evaluate", "issue next batch's primitive loads",
         "stores of this batch", "next_sample"]
tot = sum(t[:5]) + t[6] + t[7]
print(f"hit kernel: {hb} batches, {tot / hb:.0f} ticks per batch (shader clock)")
for n, v in zip(["window list (feed_next)", "issuing next batch's slot loads"] + ["waiting for this batch's slot + primitive records (vmcnt(0))", "Material::evaluate + roulette"] + names[2:], [t[6], t[0], t[7], t[1]] + t[2:5]):
    print(f"   {v / hb:8.1f} ticks  {v / tot:6.1%}  {n}")
mb_ = max(t[11], 1)
tot = sum(t[8:11])
print(f"miss kernel: {mb_} batches, {tot / mb_:.0f} ticks per batch")
for n, v in zip(["feed + issue next batch's slot loads", "wait for this batch's data + Scene::background", "next_sample"], t[8:11]):
    print(f"   {v / mb_:8.1f} ticks  {v / tot:6.1%}  {n}")
