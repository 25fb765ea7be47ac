"""The reference's example renders, for an eyeball comparison with /root/reference/examples/*.png.
The reference's pictures were made with an HDRI that is not in its repository and an unknown sample count, so they
cannot be compared number by number; these are the same scene functions (test_scenes.rs:169-274) at the reference's
native resolution (1221 x 254: 1920/500 x 400/500 cm at 125 ppi) under this build's procedural sky.
usage (GPU box): python scripts/render_examples.py <out_dir> [spp=256]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rayrs_amd
from rayrs_amd import io, procedural, scenes

out = sys.argv[1]
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 256
os.makedirs(out, exist_ok=True)
hdri = procedural.make_hdri(1024, 512)
for name in ("cook_torrance_spheres_metallic", "cook_torrance_spheres_frosted_glass", "cook_torrance_spheres_plastic",
             "material_test"):
    cam_args, objs, heur = getattr(scenes, name)()
    scene = rayrs_amd.Scene(objs, 1e-6, 1e6, heur, hdri, device=0)
    cam = rayrs_amd.Camera(*cam_args)
    chunk = rayrs_amd.frame_sample_chunk(cam.x_pixels(), cam.y_pixels(), spp)
    img, st = rayrs_amd.render(scene, cam, spp, 50, seed=0x5EED, sample_chunk=chunk)
    b, counts = io.to_raw_bytes(img)
    io.save_png(os.path.join(out, name.replace("cook_torrance_", "") + ".png"), b)
    print(f"{name}: {cam.x_pixels()}x{cam.y_pixels()} at {spp} spp, {st['rays'] / 1e6:.0f} M rays in {st['total_ms']:.1f} ms, "
          f"clamped/NaN/negative pixels {counts}", flush=True)
