"""Film sharding across GPUs: scene replicated, rank r of G renders film rows
r, r+G, r+2G, ... into a full-size framebuffer; one reduce(SUM) to rank 0 assembles the image.

Rows are interleaved (not contiguous bands) because scene content concentrates work in a few rows
(diamond_scene: the diamonds and their caustics). The sum is exact: rows a rank does not own are 0.
"""
import numpy as np


def shard_rows(rank, world, height):
    """Film rows owned by `rank` of `world`."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return np.arange(rank, height, world)


def shard_settings(rank, world):
    """(row_offset, row_stride) for igd_render_settings / RuntimeOptions.RowOffset, RowStride."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return rank, world


def reduce_framebuffer(fb, dist, dst=0):
    """The only collective of the path: SUM-reduce the per-rank framebuffers (a torch tensor on the
    backend's device: cuda for nccl/RCCL, cpu for gloo) to rank `dst`, in place."""
    dist.reduce(fb, dst=dst, op=dist.ReduceOp.SUM)
    return fb


def check_shard(fb, rank, world):
    """True when `fb` [H, W, 3] is zero outside the rows owned by `rank`."""
    h = fb.shape[0]
    mask = np.ones(h, dtype=bool)
    mask[shard_rows(rank, world, h)] = False
    return not np.asarray(fb)[mask].any()
