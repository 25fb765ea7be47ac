"""Traversal kernel: where a wave's time goes -- interior steps, leaf steps, refills (retire + new rays) -- from the
counting build's shader-clock ticks (development aid; the counting build spills, so shares are indicative).
usage: python scripts/ubench/trav_phases.py <config> <res> <spp> [lab settings "k=v,k=v"]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import rayrs_amd
from rayrs_amd import scenes, procedural
cfg, res, spp = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
cam_args, objs, heur, _, mb = scenes.config(cfg)
scene = rayrs_amd.Scene(objs, 1e-6, 1e6, heur, procedural.make_hdri(1024, 512), device=0)
cam = rayrs_amd.Camera(*scenes.camera_for_resolution(cam_args, res, res))
chunk = rayrs_amd.frame_sample_chunk(res, res, spp)
scene.lab_set(**{k: int(v, 0) for k, v in (kv.split("=") for kv in (sys.argv[4] if len(sys.argv) > 4 else "").split(",") if kv)})
rayrs_amd.render(scene, cam, 4, mb)
_, t = rayrs_amd.render(scene, cam, spp, mb, sample_chunk=chunk)
_, s = rayrs_amd.render(scene, cam, spp, mb, sample_chunk=chunk, count_work=True)
tk = s["interior_ticks"] + s["leaf_ticks"] + s["refill_ticks"]
n_int, n_leaf = s["step_wave"] // 64, s["leaf_wave"] // 64  # (the *_wave counters count 64 per wave step)
print(f"timed trav {t['kernel_ms']:.1f} ms; counting build {s['kernel_ms']:.1f} ms; rays {s['rays']/1e9:.3f} G")
print(f"interior: {s['interior_ticks']/tk:.3f} of the wave time, {n_int/1e6:.1f} M wave steps, {s['interior_ticks']/max(n_int,1):.0f} ticks each, lanes {s['step_lane']/max(s['step_wave'],1):.3f}")
print(f"leaf:     {s['leaf_ticks']/tk:.3f} of the wave time, {n_leaf/1e6:.1f} M wave steps, {s['leaf_ticks']/max(n_leaf,1):.0f} ticks each, lanes {s['inner_wave']/max(s['leaf_wave'],1):.3f}")
if s["hot_group"]:
    print(f"pre-test (by the kernels that make the rays): {s['pre_rays']/s['rays']:.3f} of the queries answered there; "
          f"{s['hot_lane']/s['rays']:.3f} of the rays put to the hot group's gate, {s['hot_prim_tests']/s['rays']:.2f} primitive tests per ray there, "
          f"{s['hot_tri_divided']/max(s['hot_prim_tests'],1):.4f} of them with the divisions made")
print(f"refill:   {s['refill_ticks']/tk:.3f} of the wave time; rays per wave-step of either kind {s['rays']/(n_int+n_leaf):.2f}")
print(f"records/ray {s['interior_visits']/s['rays']:.2f} prims/ray {(s['tri_tests']+s['sphere_tests']+s['plane_tests'])/s['rays']:.2f} leaf steps/ray {s['inner_wave']/s['rays']:.2f}")
