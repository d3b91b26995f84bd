"""Crop-level data parallelism: one process per GPU, crops sharded round-robin, no collective on the data path.

The reference has no distributed execution at all (a single process refines one crop at a time,
pipelines/refine_css.py:65,94).  Crops are fully independent -- own parameters, targets and surfels, shared read-only decoder
weights -- so the MI355X-native extension is the simplest possible one: crop i runs on rank i mod N, and the only exchange is ONE
all_gather of the small per-crop result rows (loss, yaw, t, scale, latent ... a few floats per crop) after the refinement loop,
over RCCL/xGMI on GPUs (backend "nccl") or gloo on CPU tensors (tests).  The payload is latency- not link-bound.
"""
import torch
import torch.distributed as dist


def shard_crops(n_crops, rank, world):
    """Indices of the crops owned by `rank`: i with i mod world == rank (ascending)."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world %d" % (rank, world))
    return list(range(rank, n_crops, world))


def gather_crop_results(local_rows, n_crops, rank=None, world=None, group=None):
    """All ranks receive the (n_crops, R) table of per-crop result rows in crop order.

    local_rows: (len(shard_crops(n_crops, rank, world)), R) tensor, row j belonging to crop rank + j*world.  One all_gather of
    equally padded blocks (ranks may own one crop fewer than others)."""
    if world is None:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank(group) if dist.is_initialized() else 0
    mine = shard_crops(n_crops, rank, world)
    if local_rows.dim() != 2 or local_rows.shape[0] != len(mine):
        raise ValueError("local_rows must be (%d, R) on rank %d" % (len(mine), rank))
    R = local_rows.shape[1]
    if world == 1:
        return local_rows.clone()
    per = (n_crops + world - 1) // world
    block = local_rows.new_zeros((per, R))
    block[:len(mine)] = local_rows
    blocks = [torch.empty_like(block) for _ in range(world)]
    dist.all_gather(blocks, block.contiguous(), group=group)
    out = local_rows.new_zeros((n_crops, R))
    for r in range(world):
        idx = shard_crops(n_crops, r, world)
        out[idx] = blocks[r][:len(idx)]
    return out
