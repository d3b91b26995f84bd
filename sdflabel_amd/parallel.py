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


def refine_sharded(refiner, params, nocs_pred, lidars, iters, rank=0, world=1, group=None, gather=True, K=None, crop_sizes=None, timing=None):
    """BASELINE configs[3]: refine `n_crops` independent crops sharded over the ranks.  Crop i belongs to rank i mod world; each rank refines
    its crops in chunks of `refiner.B` (sdflabel_amd.BatchRefiner, or any object with B, L, set_crops, optimize, results; or a LIST of such
    refiners: that many chunks in flight, each on its own stream) for `iters`
    iterations; ONE all_gather of the per-crop rows at the end is the only collective (pipelines/refine_css.py:65,94 loops over the same crops
    one at a time in one process).

    params     {'yaw' (n,), 'trans' (n,3), 'scale' (n,), 'latent' (n,L)} arrays for ALL crops (a few floats per crop: every rank holds the table)
    nocs_pred  (n,3,h,w) per-crop CSS predictions, or (1,3,h,w) shared by all crops (synthetic workloads)
    lidars     list of n (M_i,3) arrays, or ONE (M,3) array shared by all crops
    K, crop_sizes  (ragged refiners, BatchRefiner(max_pixels=...)): per-crop intrinsics (n,3,3) and image sizes [(H_i, W_i)] * n -- crops as the
               reference pipeline produces them (utils/refinement.py:586-609); nocs_pred is then a list of n (3,h_i,w_i) predictions
    timing     optional dict: receives this rank's seconds in 'set_crops', 'optimize' (enqueue + GPU, synchronised per chunk by results()) and
               'all_gather' (r06: lets a multi-GPU curve be read -- which phase stops scaling)
    Returns the (n_crops, 7+L) table [yaw, trans(3), scale, latent(L), weighted 2-D loss, weighted 3-D loss] -- the refined parameters and the
    per-crop losses of the last iteration (BASELINE.json north_star: "all-gather of the per-crop losses") -- in crop order on every rank
    (gather=False: this rank's rows only).
    A short last chunk is padded with copies of its last crop (the padded rows are dropped).  If the local refinement fails, the rank still
    takes part in the collective (NaN rows) and raises afterwards, so no rank is left waiting."""
    import contextlib
    import numpy as np
    n_crops = int(np.asarray(params["yaw"]).reshape(-1).shape[0])
    mine = shard_crops(n_crops, rank, world)
    # r06: `refiner` may be a LIST of refiners of one batch size: this rank's chunks are then refined that many at a time, each refiner's iterations
    # replayed on a stream of its own -- the matrix-core-bound decoder passes of one chunk run beside the VALU-bound splat / loss kernels of the
    # other (+10 % crops/s with two refiners of 64 crops and the float16 decoder on one MI355X, tools/two_stream_ab.py; results unchanged: the
    # chunks are independent)
    refiners = list(refiner) if isinstance(refiner, (list, tuple)) else [refiner]
    refiner = refiners[0]
    B, R = int(refiner.B), 7 + int(refiner.L)
    if any(int(r.B) != B or int(r.L) != int(refiner.L) for r in refiners):
        raise ValueError("refine_sharded: the refiners of a list must share batch size and latent size")
    P = {k: np.asarray(v, np.float32).reshape(n_crops, -1) for k, v in params.items()}
    import time
    tm = {"set_crops": 0.0, "optimize": 0.0, "all_gather": 0.0, "chunks": 0, "refiners_in_flight": len(refiners)}
    ragged = K is not None or crop_sizes is not None
    shared_target = (not isinstance(nocs_pred, (list, tuple))) and nocs_pred.shape[0] == 1
    shared_lidar = not isinstance(lidars, (list, tuple))
    rows, failure = [], None

    def load(rf, ids):
        n = len(ids)
        sel = ids + [ids[-1]] * (B - n)
        if isinstance(nocs_pred, (list, tuple)):
            tgt = [nocs_pred[i] for i in sel]
        else:
            tgt = nocs_pred.expand(B, *nocs_pred.shape[1:]) if shared_target else nocs_pred[sel]
        extra = {}
        if ragged:
            extra = {"K": None if K is None else np.asarray(K, np.float32).reshape(n_crops, 3, 3)[sel],
                     "crop_sizes": None if crop_sizes is None else [tuple(crop_sizes[i]) for i in sel]}
        rf.set_crops({k: v[sel] for k, v in P.items()}, tgt, [lidars] * B if shared_lidar else [lidars[i] for i in sel], **extra)

    def collect(rf, n):
        res, l2, l3 = rf.results()                       # (synchronises: the chunk's GPU time lands in 'optimize')
        z = res.new_zeros((res.shape[0], 1))
        rows.append(torch.cat([res, z if l2 is None else l2.reshape(-1, 1).to(res), z if l3 is None else l3.reshape(-1, 1).to(res)], 1)[:n])

    try:
        chunks = [mine[c0:c0 + B] for c0 in range(0, len(mine), B)]
        on_gpu = len(refiners) > 1 and all(getattr(getattr(r, "dev", None), "type", "cpu") == "cuda" for r in refiners)
        streams = [torch.cuda.Stream(device=r.dev) for r in refiners] if on_gpu else [None] * len(refiners)
        for g0 in range(0, len(chunks), len(refiners)):
            flight = list(zip(refiners, streams, chunks[g0:g0 + len(refiners)]))
            t0 = time.perf_counter()
            for rf, _, ids in flight:
                load(rf, ids)
            t1 = time.perf_counter()
            if len(flight) == 1:
                flight[0][0].optimize(iters)
            else:
                cur = torch.cuda.current_stream(refiner.dev) if on_gpu else None
                for _, st, _ in flight:
                    if st is not None:
                        st.wait_stream(cur)
                for _ in range(iters):                   # iteration by iteration, so that the refiners' launches interleave on the device
                    for rf, st, _ in flight:
                        with (torch.cuda.stream(st) if st is not None else contextlib.nullcontext()):
                            rf.optimize(1)
                for _, st, _ in flight:
                    if st is not None:
                        cur.wait_stream(st)
            for rf, _, ids in flight:
                collect(rf, len(ids))
            tm["set_crops"] += t1 - t0
            tm["optimize"] += time.perf_counter() - t1
            tm["chunks"] += len(flight)
        local = torch.cat(rows) if rows else None
    except Exception as e:                                   # noqa: BLE001 -- re-raised below, after the collective
        failure, local = e, None
    dev = getattr(refiner, "dev", "cpu")
    if local is None:
        local = torch.full((len(mine), R), float("nan") if failure is not None else 0.0, dtype=torch.float32, device=dev)
    t0 = time.perf_counter()
    table = gather_crop_results(local, n_crops, rank, world, group) if gather else local
    tm["all_gather"] = time.perf_counter() - t0
    if timing is not None:
        timing.update(tm)
    if failure is not None:
        raise failure
    return table
