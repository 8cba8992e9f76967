"""Sharding across GPUs: scene replicated, no data-path exchange, one reduce(SUM) at the end.

Two partitions of the same unit (camera samples), both exact up to float summation order:
  * rows:       rank r of G renders film rows r, r+G, r+2G, ... of every iteration (strong scaling of one image)
  * iterations: rank r renders the whole-film iterations r*K .. r*K + K - 1 of G*K (the per-sample RNG depends on the iteration
                index, core/random.art:34-43, so the union is the single-device sample set; per-GPU work stays a
                full iteration however many GPUs take part — the partition to use when the image is small)

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


def shard_iterations(rank, world, steps):
    """Global iteration indices rendered by `rank` when every rank runs `steps` steps: a contiguous block, so that the
    device can execute a rank's consecutive iterations as one wavefront (igd_device.h, deferred batching)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return [rank * steps + i for i in range(steps)]


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
