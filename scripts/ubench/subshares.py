"""One rank's share of a frame rendered as n concurrent sub-shares: thread i renders the tiles t % (ranks * n) == rank + ranks * i
on its own clone of the scene and its own stream, all into the same framebuffer; a frame is done when all n are (no overlap
between frames).  usage: python scripts/ubench/subshares.py <config> <res> <spp> <frames> <ranks>"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import rayrs_amd
from rayrs_amd import scenes, procedural, api

cfg, res, spp, frames, ranks = (int(a) for a in sys.argv[1:6])
cam_args, objs, heur, _, mb = scenes.config(cfg)
cam_args = scenes.camera_for_resolution(cam_args, res, res)
scene = rayrs_amd.Scene(objs, 1e-6, 1e6, heur, procedural.make_hdri(1024, 512), device=0)
cam = rayrs_amd.Camera(*cam_args)
chunk = rayrs_amd.frame_sample_chunk(res, res, spp)
handles = [scene] + [scene.clone_to_device(0) for _ in range(3)]
streams = [torch.cuda.Stream(device=0) for _ in handles]
fb = torch.zeros((res, res, 3), dtype=torch.float32, device="cuda:0")
ref = None
for n in (1, 2, 3, 4, 2, 1):
    params = [api.make_params(spp, mb, 0x5EED, chunk, 0 + ranks * i, ranks * n, False, False) for i in range(n)]

    def frame():
        def work(i):
            api.render_launch(handles[i], cam, params[i], fb.data_ptr(), streams[i].cuda_stream)
            api.render_finish(handles[i])
        th = [threading.Thread(target=work, args=(i,)) for i in range(n)]
        for t in th:
            t.start()
        for t in th:
            t.join()

    fb.zero_()
    frame()  # warm
    torch.cuda.synchronize()
    fb.zero_()
    t0 = time.perf_counter()
    for _ in range(frames):
        frame()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    img = fb.cpu().numpy()
    if ref is None:
        ref = img.copy()
    print(f"{n} sub-shares of rank 0 of {ranks}: {dt * 1e3 / frames:7.1f} ms per frame  same_bits={bool((img.view('u4') == ref.view('u4')).all())}", flush=True)
