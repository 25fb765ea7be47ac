"""Image-tile sharding across the GPUs of one node.

Paths never communicate (rayrs/src/main.rs:61-79) and the RNG is keyed by
(pixel, sample), so any partition of the image gives the same pixels.  Rank r
renders the 8x8 tiles whose row-major index t satisfies t % ranks == r
(rayrs_render_params.tile_rank / tile_ranks) into a zeroed full-size f32x3
framebuffer; one RCCL reduce (sum) to rank 0 over xGMI assembles the frame.  The
other ranks hold exact zeros for a pixel, so the sum is exact and independent of
the reduction order: the N-GPU frame is bit-identical to the 1-GPU frame.
"""
import numpy as np

TILE = 8


def tile_mask(width: int, height: int, rank: int, ranks: int) -> np.ndarray:
    """Boolean (height, width) mask of the pixels rank `rank` of `ranks` owns."""
    tiles_x = (width + TILE - 1) // TILE
    rows = np.arange(height)[:, None] // TILE
    cols = np.arange(width)[None, :] // TILE
    return ((rows * tiles_x + cols) % ranks) == rank


def local_tile_count(width: int, height: int, rank: int, ranks: int) -> int:
    n = ((width + TILE - 1) // TILE) * ((height + TILE - 1) // TILE)
    return (n - rank + ranks - 1) // ranks if n > rank else 0


def reduce_framebuffer(fb, dst: int = 0):
    """Sum the per-rank framebuffers into rank `dst` (torch.distributed; 'nccl' is RCCL on ROCm)."""
    import torch.distributed as dist
    dist.reduce(fb, dst=dst, op=dist.ReduceOp.SUM)
    return fb
