"""Local-pool kernel: where a wave's time goes, per phase kind (count_work build; development aid).
usage: python scripts/ubench/lp_phases.py <config> <res> <spp>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import rayrs_amd
from rayrs_amd import scenes, procedural
cfg, res, spp = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
cam_args, objs, heur, _, mb = scenes.config(cfg)
scene = rayrs_amd.Scene(objs, 1e-6, 1e6, heur, procedural.make_hdri(1024, 512), device=0)
cam = rayrs_amd.Camera(*scenes.camera_for_resolution(cam_args, res, res))
chunk = rayrs_amd.frame_sample_chunk(res, res, spp)
rayrs_amd.render(scene, cam, 4, mb)
_, t = rayrs_amd.render(scene, cam, spp, mb, sample_chunk=chunk)
_, s = rayrs_amd.render(scene, cam, spp, mb, sample_chunk=chunk, count_work=True)
hits = sum(s["surface_hits"])
print(f"timed {t['kernel_ms']:.2f} ms ({t['rays'] / t['kernel_ms'] / 1e3:.0f} Mray/s); counting build {s['kernel_ms']:.2f} ms")
print(f"rays {s['rays']/1e6:.1f} M, hits {hits/1e6:.1f} M, escaped {s['escaped_paths']/1e6:.1f} M, paths {s['paths']/1e6:.1f} M; lane utilisation {s['step_lane']/s['step_wave']:.3f}; phase executions {s['step_wave']/64e6:.2f} M")
import numpy as np
out = np.zeros(10, dtype=np.uint64)
rayrs_amd._ffi.lib().rayrs_debug_counters(scene._h, out.ctypes.data)
tot = float(out[1:5].sum())
# (ISECT: the query at the end of a GEN / BG+GEN / SHADE_k execution; the counting build waits for memory at every stamp)
for k, name in ((1, "GEN"), (2, "ISECT"), (3, "BG"), (4, "SHADE")):
    print(f"  {name:6s} {out[k] / tot:6.3f} of the wave time, {out[5 + k] / 1e6:8.2f} M executions, {out[k] / max(int(out[5 + k]), 1):8.0f} ticks each")
