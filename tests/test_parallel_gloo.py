"""world_size-2 gloo test (CPU) of the crop sharding / result gathering used by bench.py for N > 1."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sdflabel_amd.parallel import gather_crop_results, shard_crops


def _crop_result(i):
    return torch.tensor([i, 0.1 * i, -2.0 * i, i * i], dtype=torch.float32)


def _worker(rank, world, port, n_crops, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine = shard_crops(n_crops, rank, world)
        rows = torch.stack([_crop_result(i) for i in mine]) if mine else torch.zeros((0, 4))
        table = gather_crop_results(rows, n_crops)
        q.put((rank, mine, table.clone()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_is_a_partition():
    for n in (0, 1, 7, 8, 1024):
        for w in (1, 2, 8):
            parts = [shard_crops(n, r, w) for r in range(w)]
            assert sorted(sum(parts, [])) == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    with pytest.raises(ValueError):
        shard_crops(4, 2, 2)


@pytest.mark.parametrize("world,n_crops", [(2, 2), (2, 5), (8, 1021), (8, 5)])
def test_gather_ranks_gloo(world, n_crops):
    """world 2 and the 8-rank layout of BASELINE configs[3] with an uneven split (1021 crops: five ranks own 128, three own 127) and a
    split with idle ranks (5 crops on 8 ranks)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_crops, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = torch.stack([_crop_result(i) for i in range(n_crops)])
    for rank, mine, table in got:
        assert mine == list(range(rank, n_crops, world))
        assert torch.equal(table, ref)


def test_single_process_passthrough():
    rows = torch.stack([_crop_result(i) for i in range(3)])
    assert torch.equal(gather_crop_results(rows, 3, rank=0, world=1), rows)
