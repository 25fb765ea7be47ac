import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import rayrs_amd
from rayrs_amd import scenes, procedural
cam_args, objs, heur, _, mb = scenes.config(5)
hdri = procedural.make_hdri(1024, 512)
scene = rayrs_amd.Scene(objs, 1e-6, 1e6, heur, hdri, device=0)
for name, o, l in (("reference", cam_args[0], cam_args[2]), ("horizon", (0.0, 6.0, 10.0), (0.0, 5.5, 0.0)), ("sky", (0.0, 5.0, 10.0), (0.0, 9.0, 20.0))):
    ca = scenes.camera_for_resolution((o, cam_args[1], l) + tuple(cam_args[3:]), 2048, 2048)
    cam = rayrs_amd.Camera(*ca)
    rayrs_amd.render(scene, cam, 4, mb, sample_chunk=4)
    img, st = rayrs_amd.render(scene, cam, 256, mb, sample_chunk=4)
    print(f"{name}: {st['rays']/st['trace_ms']/1e3:.1f} Mray/s, {st['trace_ms']:.1f} ms, rays {st['rays']}, escaped {st['escaped_paths']}, rounds {st['kernel_launches']}", flush=True)
