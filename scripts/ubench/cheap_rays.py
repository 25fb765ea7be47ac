"""What the streaming kernels spend on queries that never come near the mesh (development aid): the headline
scene from a camera that sees only floor and sky, against the reference framing.  Prints per-kernel-class times."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import rayrs_amd
from rayrs_amd import scenes, procedural

res, spp = int(sys.argv[1]), int(sys.argv[2])
cam_args, objs, heur, _, mb = scenes.config(5)
scene = rayrs_amd.Scene(objs, 1e-6, 1e6, heur, procedural.make_hdri(1024, 512), device=0)
chunk = rayrs_amd.frame_sample_chunk(res, res, spp)
for name, ca in (("reference", cam_args),
                 ("floor only", ((15., 5., 15.), (0., 1., 0.), (20., 1., 20.), 50., 1., 1., 100)),
                 ("close", scenes.MESH_CLOSE_CAM)):
    cam = rayrs_amd.Camera(*scenes.camera_for_resolution(ca, res, res))
    rayrs_amd.render(scene, cam, 4, mb)
    for count in (False, True):
        img, st = rayrs_amd.render(scene, cam, spp, mb, sample_chunk=chunk, count_work=count)
        prims = st['tri_tests'] + st['sphere_tests'] + st['plane_tests']
        print(f"{name:10s} count={count} rays {st['rays']/1e6:8.1f} M  trace {st['trace_ms']:8.1f} ms  trav {st['kernel_ms']:8.1f} ms  "
              f"rounds {st['kernel_launches']}  Mray/s {st['rays']/st['trace_ms']/1e3:8.1f}  trav ns/ray {st['kernel_ms']*1e6/st['rays']:.4f}"
              + (f"  rec/ray {st['interior_visits']/st['rays']:.2f} prims/ray {prims/st['rays']:.2f} direct {st['direct_rays']/st['rays']:.3f}" if count else ""), flush=True)
