"""Quick GPU throughput probe (development aid, not the benchmark)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rayrs_amd
from rayrs_amd import scenes, procedural

def run(cfg, w, h, spp, chunk=0, level=None, count=False, ranks=1):
    if level is not None:
        cam_args, objs, heur = scenes.mesh_scene(level)
        cam_args = scenes.camera_for_resolution(cam_args, w, h); mb = 50
    else:
        cam_args, objs, heur, _, mb = scenes.config(cfg)
        cam_args = scenes.camera_for_resolution(cam_args, w, h)
    hdri = procedural.make_hdri(1024, 512)
    t = time.time()
    scene = rayrs_amd.Scene(objs, 1e-6, 1e6, heur, hdri, device=0)
    tune = {k: int(v) for k, v in (kv.split("=") for kv in os.environ.get("PROBE_TUNING", "").split(",") if kv)}
    if tune:
        public = {k: v for k, v in tune.items() if k in ("pool_slots", "local_pool")}
        scene.set_tuning(**public)
        scene.lab_set(**{k: v for k, v in tune.items() if k not in public})
    tb = time.time() - t
    cam = rayrs_amd.Camera(*cam_args)
    info = scene.info()
    img, st = rayrs_amd.render(scene, cam, min(spp, 4), mb, sample_chunk=chunk)  # warm
    img, st = rayrs_amd.render(scene, cam, spp, mb, sample_chunk=chunk, count_work=count, tile_ranks=ranks)
    mr = st['rays'] / st['trace_ms'] / 1e3
    print(f"cfg={cfg} {w}x{h}x{spp} chunk={chunk} ranks={ranks} Mray/s={mr:.1f} trace={st['trace_ms']:.1f}ms trav={st['kernel_ms']:.1f}ms rounds={st['kernel_launches']} rays={st['rays']}", flush=True)
    if count:
        prims = st['tri_tests'] + st['sphere_tests'] + st['plane_tests']
        print(f"  visits/ray={st['interior_visits']/st['rays']:.1f} prims/ray={prims/st['rays']:.2f} | lane utilisation: "
              f"interior {st['step_lane']/max(st['step_wave'],1)/1:.2f} leaf {st['inner_wave']/max(st['leaf_wave'],1):.2f} "
              f"| wave-phases/ray*64: int {st['step_wave']/st['rays']:.2f} leaf {st['leaf_wave']/st['rays']:.2f}", flush=True)
        tk = st['interior_ticks'] + st['leaf_ticks'] + st['refill_ticks']
        print(f"  wave time: interior {st['interior_ticks']/tk:.2f} leaf {st['leaf_ticks']/tk:.2f} refill {st['refill_ticks']/tk:.2f} | "
              f"ticks per wave-phase: interior {st['interior_ticks']/max(st['step_wave']/64,1):.0f} leaf {st['leaf_ticks']/max(st['leaf_wave']/64,1):.0f} | refill ticks per ray {st['refill_ticks']/st['rays']:.1f}", flush=True)
    return st

if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("all", "c2"):
        run(2, 1024, 1024, 256, chunk=16)
    if what in ("all", "c4"):
        run(4, 1024, 1024, 16)
    if what in ("all", "c3"):
        run(3, 1024, 1024, 256, chunk=16)
    if what in ("all", "c5"):
        run(5, 1024, 1024, 256, chunk=16)
    if what == "u5":
        run(5, 1024, 1024, 128, chunk=16, count=True)
    if what == "u2":
        run(2, 1024, 1024, 128, chunk=16, count=True)
    if what == "big4":
        run(4, 2048, 2048, 128, chunk=4)
    if what == "big5":
        run(5, 2048, 2048, 64, chunk=16)
    if what == "full5":
        run(5, 2048, 2048, 1024, chunk=4)
    if what == "shard8":
        chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 16
        run(5, 2048, 2048, 1024, chunk=chunk, ranks=8)
