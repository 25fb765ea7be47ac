"""Stream-pool kernel: lane utilisation and counters on the headline scene (development aid).
usage: python scripts/ubench/sp_probe.py <res> <spp> [tuning k=v,...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import rayrs_amd
from rayrs_amd import scenes, procedural
res, spp = int(sys.argv[1]), int(sys.argv[2])
tune = {k: int(v) for k, v in (kv.split("=") for kv in (sys.argv[3] if len(sys.argv) > 3 else "").split(",") if kv)}
cam_args, objs, heur, _, mb = scenes.config(5)
scene = rayrs_amd.Scene(objs, 1e-6, 1e6, heur, procedural.make_hdri(1024, 512), device=0)
if tune: scene.set_tuning(**tune)
cam = rayrs_amd.Camera(*scenes.camera_for_resolution(cam_args, res, res))
chunk = rayrs_amd.frame_sample_chunk(res, res, spp)
rayrs_amd.render(scene, cam, 4, mb)
_, t = rayrs_amd.render(scene, cam, spp, mb, sample_chunk=chunk)
_, s = rayrs_amd.render(scene, cam, spp, mb, sample_chunk=chunk, count_work=True)
print(f"timed: trace {t['trace_ms']:.1f} ms trav {t['kernel_ms']:.1f} shade {t['hit_ms']:.1f} rounds {t['kernel_launches']} Mray/s {t['rays']/t['trace_ms']/1e3:.0f}")
print(f"rays {s['rays']/1e6:.1f} M; stream-pool lane utilisation {s['shade_lane']/max(s['shade_wave'],1):.3f}; phase executions {s['shade_wave']/64/1e6:.2f} M "
      f"({s['shade_wave']/64/max(s['rays'],1)*64:.2f} per 64 rays)")
import ctypes as C, numpy as np
out = np.zeros(10, dtype=np.uint64)
rayrs_amd._ffi.lib().rayrs_debug_counters(scene._h, out.ctypes.data)
tot = float(out[:5].sum())
for j, name in enumerate(("IMPORT", "GEN", "ISECT", "BG", "SHADE")):
    print(f"  {name:7s} {out[j] / tot:6.3f} of the wave time, {out[5 + j] / 1e6:8.2f} M executions, {out[j] / max(int(out[5 + j]), 1):8.0f} ticks each")
