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


def gather_rows(fb, rank, world, dist, dst=0):
    """The collective of the tile-sharded path: every rank sends ONLY the film rows it owns (ceil(H / world) x W x 3 floats,
    i.e. 1 / world of the framebuffer: 25 MB per rank for a 4096 x 4096 film on 8 GPUs) and rank `dst` writes
    them into their places of its own framebuffer. `fb` is a torch tensor [H, W, 3] on the backend's device (cuda for
    nccl = RCCL, cpu for gloo) whose rows rank, rank + world, ... hold this rank's shard. In place on `dst`; other ranks keep
    their shard. Exact: no arithmetic happens on the way (a reduce(SUM) of zero-padded framebuffers gives the same image but
    moves `world` times the bytes)."""
    import torch
    h = fb.shape[0]
    rows_max = (h + world - 1) // world
    mine = fb[rank::world]
    packed = torch.zeros((rows_max,) + tuple(fb.shape[1:]), dtype=fb.dtype, device=fb.device)
    packed[:mine.shape[0]] = mine
    if world == 1:
        return fb
    if rank == dst:
        parts = [torch.empty_like(packed) for _ in range(world)]
        dist.gather(packed, parts, dst=dst)
        for r in range(world):
            if r != dst:
                n = len(range(r, h, world))
                fb[r::world] = parts[r][:n]
    else:
        dist.gather(packed, None, dst=dst)
    return fb


def gather_bytes(height, width, world):
    """Bytes each rank contributes to gather_rows."""
    return ((height + world - 1) // world) * width * 3 * 4
