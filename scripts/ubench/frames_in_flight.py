"""Frames in flight on one GPU: n host threads, each with its own clone of the scene (own path pool) and its own HIP
stream, render `frames` frames between them -- the end of one frame (a thinning pool, small kernels) overlaps the start
of the next.  usage: python scripts/ubench/frames_in_flight.py <config> <res> <spp> <frames> [tile_ranks]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import rayrs_amd
from rayrs_amd import scenes, procedural, api

cfg, res, spp, frames = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
ranks = int(sys.argv[5]) if len(sys.argv) > 5 else 1
cam_args, objs, heur, _, mb = scenes.config(cfg)
cam_args = scenes.camera_for_resolution(cam_args, res, res)
scene = rayrs_amd.Scene(objs, 1e-6, 1e6, heur, procedural.make_hdri(1024, 512), device=0)
cam = rayrs_amd.Camera(*cam_args)
chunk = rayrs_amd.frame_sample_chunk(res, res, spp)
params = api.make_params(spp, mb, 0x5EED, chunk, 0, ranks, False, False)
handles = [scene, scene.clone_to_device(0), scene.clone_to_device(0)]
streams = [torch.cuda.Stream(device=0) for _ in handles]
bufs = [torch.zeros((res, res, 3), dtype=torch.float32, device="cuda:0") for _ in handles]
ref = None
for n in (1, 2, 3, 2, 1):
    for b in bufs:
        b.zero_()
    torch.cuda.synchronize()
    rays = [0] * n

    def work(i):
        for k in range(i, frames, n):
            api.render_launch(handles[i], cam, params, bufs[i].data_ptr(), streams[i].cuda_stream)
            rays[i] += api.render_finish(handles[i])["rays"]

    for i in range(n):  # warm every clone's pool
        api.render_launch(handles[i], cam, params, bufs[i].data_ptr(), streams[i].cuda_stream)
        api.render_finish(handles[i])
    torch.cuda.synchronize()
    rays = [0] * n
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(i,)) for i in range(n)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    img = bufs[0].cpu().numpy()
    if ref is None:
        ref = img.copy()
    same = all(bool((bufs[i].cpu().numpy().view("u4") == ref.view("u4")).all()) for i in range(n))
    print(f"{n} in flight: {frames} frames in {dt * 1e3:8.1f} ms = {dt * 1e3 / frames:7.1f} ms per frame, {sum(rays) / dt / 1e6:8.1f} Mray/s  same_bits={same}", flush=True)
