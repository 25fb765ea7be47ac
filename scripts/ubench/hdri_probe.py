"""What the HDRI table's size costs the miss kernel: the headline frame with environment maps of several resolutions
(64-byte records: 2 MB at 256x128 -- resident in every L2 -- 34 MB at 1024x512, 134 MB at 2048x1024).  Development aid.
usage: python scripts/ubench/hdri_probe.py [config]
Measured: miss 211 / 215 / 222 / 225 ms at 256x128 / 512x256 / 1024x512 / 2048x1024: the table's misses cost 5 %."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import rayrs_amd
from rayrs_amd import scenes, procedural
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 5
res, spp = {5: (2048, 1024), 2: (1024, 256), 4: (2048, 512), 3: (1024, 512), 1: (256, 64)}[cfg]
cam_args, objs, heur, _, mb = scenes.config(cfg)
cam_args = scenes.camera_for_resolution(cam_args, res, res)
for (w, h) in ((1024, 512), (256, 128), (1024, 512), (512, 256), (2048, 1024)):
    scene = rayrs_amd.Scene(objs, 1e-6, 1e6, heur, procedural.make_hdri(w, h), device=0)
    cam = rayrs_amd.Camera(*cam_args)
    rayrs_amd.render(scene, cam, 4, mb, sample_chunk=0)
    img, st = rayrs_amd.render(scene, cam, spp, mb, sample_chunk=rayrs_amd.frame_sample_chunk(res, res, spp))
    print(f"hdri {w}x{h}: trace {st['trace_ms']:.1f} trav {st['kernel_ms']:.1f} hit {st['hit_ms']:.1f} miss {st['miss_ms']:.1f} rays {st['rays']}", flush=True)
    del scene
