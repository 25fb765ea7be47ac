"""The N>1 path on CPU: world_size-2, -3 and -8 gloo processes each hold the tiles they own
(zeros elsewhere) and one reduce assembles the frame on rank 0, exactly.  (Eight, a full node's rank
count, can only be rehearsed here: the GPU pool admits six processes on a card.)"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def test_tile_masks_partition_the_image():
    from rayrs_amd import tiles
    for (w, h) in ((64, 48), (61, 19), (2048, 2048)):
        for ranks in (1, 2, 3, 8):
            masks = [tiles.tile_mask(w, h, r, ranks) for r in range(ranks)]
            total = np.zeros((h, w), dtype=np.int32)
            for m in masks:
                total += m
            assert np.all(total == 1)
            counts = [tiles.local_tile_count(w, h, r, ranks) for r in range(ranks)]
            assert sum(counts) == ((w + 7) // 8) * ((h + 7) // 8)
            assert max(counts) - min(counts) <= 1


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import _oracle
    from rayrs_amd import procedural, scenes, tiles
    cam_args, objs, heur = scenes.material_test()
    cam_args = scenes.camera_for_resolution(cam_args, 61, 19)  # ragged: not a multiple of the tile size
    hdri = procedural.make_hdri(64, 32)
    osc = _oracle.OracleScene(objs, 1e-6, 1e6, heur, hdri)
    ocam = _oracle.OracleCamera(*cam_args)
    full, _ = osc.render(ocam, 4, 50, seed=3, nthreads=1)
    full32 = full.astype(np.float32)                      # image.rs:224-229
    mine = np.where(tiles.tile_mask(61, 19, rank, world)[:, :, None], full32, np.float32(0))
    fb = torch.from_numpy(np.ascontiguousarray(mine))
    tiles.reduce_framebuffer(fb, dst=0)
    dist.barrier()
    if rank == 0:
        np.save(out_path, np.stack([fb.numpy(), full32]))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_gloo_reduce_assembles_the_exact_frame(tmp_path, world):
    out = str(tmp_path / "frame.npy")
    port = 29500 + (os.getpid() % 2000) + world
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    got, want = np.load(out)
    assert np.array_equal(got, want)
