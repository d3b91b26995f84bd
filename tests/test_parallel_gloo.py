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
        q.put((rank, mine, table.numpy().copy()))        # by value: a tensor travels as a file descriptor the exiting worker may close first
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
        assert torch.equal(torch.from_numpy(table), ref)


def test_single_process_passthrough():
    rows = torch.stack([_crop_result(i) for i in range(3)])
    assert torch.equal(gather_crop_results(rows, 3, rank=0, world=1), rows)


# ---- refine_sharded: the function bench.py's configs[3] sections call, with a stand-in refiner on CPU ranks -------------------------------

class _FakeRefiner:
    """duck-typed stand-in for sdflabel_amd.BatchRefiner on CPU: `optimize(k)` adds k to every parameter, so that the result of crop i is a
    known function of its initial row and the chunking / padding / gathering logic is what is tested"""

    def __init__(self, B, L=3):
        self.B, self.L, self.dev = B, L, torch.device("cpu")
        self.calls = []

    def set_crops(self, params, nocs_pred, lidars):
        assert all(v.shape[0] == self.B for v in params.values()) and nocs_pred.shape[0] == self.B and len(lidars) == self.B
        self.rows = torch.cat([torch.as_tensor(params[k], dtype=torch.float32).reshape(self.B, -1) for k in ("yaw", "trans", "scale", "latent")], 1)
        self.calls.append(self.rows[:, 0].clone())

    def optimize(self, iters):
        self.rows = self.rows + float(iters)

    def results(self):
        return self.rows.clone(), None, None


def _params(n):
    import numpy as np
    i = np.arange(n, dtype=np.float32)
    return {"yaw": i, "trans": np.stack([i, 2 * i, 3 * i], 1), "scale": np.full(n, 2.0, np.float32), "latent": np.stack([-i, i * 0.5, i + 0.25], 1)}


def _expected(n, iters):
    p = _params(n)
    rows = torch.cat([torch.from_numpy(p[k]).reshape(n, -1) for k in ("yaw", "trans", "scale", "latent")], 1) + float(iters)
    return torch.cat([rows, torch.zeros(n, 2)], 1)           # + the two per-crop loss columns (the stand-in refiner reports none)


def _sharded_worker(rank, world, port, n_crops, chunk, fail_rank, q):
    from sdflabel_amd.parallel import refine_sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rf = _FakeRefiner(chunk)
        if rank == fail_rank:
            rf.optimize = lambda iters: (_ for _ in ()).throw(RuntimeError("boom on rank %d" % rank))
        err, table = None, None
        try:
            table = refine_sharded(rf, _params(n_crops), torch.zeros(1, 3, 4, 4), torch.zeros(5, 3), 7, rank, world)
        except RuntimeError as e:
            err = str(e)
        q.put((rank, None if table is None else table.numpy().copy(), err, len(rf.calls)))     # (by value, see above)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_crops,chunk", [(2, 10, 4), (2, 7, 64), (8, 1021, 64), (8, 5, 2)])
def test_refine_sharded_chunks_pads_and_gathers_gloo(world, n_crops, chunk):
    """crop i -> rank i mod world, chunks of `chunk` with the short last chunk padded, one all_gather: every rank ends with the full table"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_worker, args=(r, world, port, n_crops, chunk, -1, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = _expected(n_crops, 7)
    for rank, table, err, calls in got:
        assert err is None and torch.equal(torch.from_numpy(table), ref), rank
        mine = len(range(rank, n_crops, world))
        assert calls == (mine + chunk - 1) // chunk


def test_refine_sharded_local_failure_still_joins_the_collective_gloo():
    """a rank whose refinement raises contributes NaN rows to the all_gather and re-raises afterwards; the other rank is not left waiting"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, 6, 2, 1, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {r: (t, e) for r, t, e, _ in [q.get(timeout=120) for _ in range(2)]}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[1][1] is not None and "boom" in got[1][1]
    t0 = torch.from_numpy(got[0][0])
    assert got[0][1] is None and torch.equal(t0[0::2], _expected(6, 7)[0::2]) and bool(torch.isnan(t0[1::2]).all())


def test_refine_sharded_single_rank_without_a_process_group():
    from sdflabel_amd.parallel import refine_sharded
    table = refine_sharded(_FakeRefiner(4), _params(9), torch.zeros(9, 3, 4, 4), [torch.zeros(2, 3)] * 9, 3)
    assert torch.equal(table, _expected(9, 3))


# ---- bench.py --gpus N started WITHOUT a launcher must become N ranks (or refuse) ---------------------------------------------------------

def _bench(*argv, timeout=240):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "SDFR_SELF_LAUNCHED")}
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + list(argv), capture_output=True, text=True, timeout=timeout, env=env)


def test_bench_self_launches_two_ranks_when_started_without_a_launcher():
    """plain `python bench.py --gpus 2 --launch-check`: the script re-launches itself under torch.distributed.run, both ranks join the
    process group (gloo here: no GPU in this container; RCCL on the GPU box) and rank 0 reports the group's size"""
    import json
    r = _bench("--gpus", "2", "--launch-check")
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["launch_check"] and line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["ranks"] == [0, 1]
    assert line["launcher"].startswith("self")


def test_bench_line_of_a_two_rank_job_is_compact_and_carries_the_rank_count():
    """`--launch-check --line-check`: two gloo ranks, rank 0 prints the line a real run prints (canned sections): parseable, < 4 KB, rccl_ranks 2"""
    import json
    r = _bench("--gpus", "2", "--launch-check", "--line-check")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 alone prints, and one line only"
    assert len(lines[0].encode()) < 4096
    line = json.loads(lines[0])
    assert line["rccl_ranks"] == 2 and line["n_gpus"] == 2 and line["metric"].startswith("rendered rays/sec") and "roofline" in line


def test_bench_refuses_more_ranks_than_gpus_instead_of_measuring_one():
    if torch.cuda.is_available() and torch.cuda.device_count() >= 8:
        pytest.skip("this node has 8 GPUs")
    r = _bench("--gpus", "8", "--steps", "2", "--warmup", "1", "--no-extras", "--no-cpu-baseline")
    assert r.returncode != 0 and "refusing" in (r.stderr + r.stdout)
    assert not any(l.startswith("{") for l in r.stdout.splitlines()), "no JSON line may be printed by fewer ranks than --gpus"


def test_bench_rejects_a_rank_count_that_disagrees_with_gpus():
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--launch-check"], capture_output=True, text=True, timeout=240, env=env)
    assert r.returncode != 0 and "must agree" in (r.stderr + r.stdout)


def test_refine_sharded_ragged_crops_and_phase_timing():
    """r06: crops as the pipeline produces them -- a LIST of per-crop predictions, per-crop K and (H, W) -- reach set_crops chunk by chunk in crop
    order (the short last chunk padded with its last crop), and the per-rank phase seconds are reported"""
    import numpy as np
    from sdflabel_amd.parallel import refine_sharded
    n, B = 7, 4
    seen = []

    class _Ragged(_FakeRefiner):
        def set_crops(self, params, nocs_pred, lidars, K=None, crop_sizes=None):
            assert isinstance(nocs_pred, list) and len(nocs_pred) == self.B and K.shape == (self.B, 3, 3) and len(crop_sizes) == self.B
            for t, (h, w) in zip(nocs_pred, crop_sizes):
                assert tuple(t.shape) == (3, h, w)
            seen.append(([int(k[0, 2]) for k in K], list(crop_sizes)))
            self.rows = torch.cat([torch.as_tensor(params[k], dtype=torch.float32).reshape(self.B, -1) for k in ("yaw", "trans", "scale", "latent")], 1)

    sizes = [(3 + i, 5 + 2 * i) for i in range(n)]
    Ks = np.stack([np.array([[10, 0, i], [0, 10, 0], [0, 0, 1]], np.float32) for i in range(n)])
    tm = {}
    table = refine_sharded(_Ragged(B), _params(n), [torch.zeros(3, h, w) for h, w in sizes], [torch.zeros(2, 3)] * n, 3, K=Ks, crop_sizes=sizes, timing=tm)
    assert torch.equal(table, _expected(n, 3))
    assert seen[0][0] == [0, 1, 2, 3] and seen[1][0] == [4, 5, 6, 6] and seen[1][1] == [sizes[4], sizes[5], sizes[6], sizes[6]]
    assert tm["chunks"] == 2 and all(tm[k] >= 0.0 for k in ("set_crops", "optimize", "all_gather"))


def test_refine_sharded_with_two_refiners_in_flight_gives_the_same_table():
    """r06: a LIST of refiners = that many chunks in flight (each on a stream of its own on the GPU; in turn on CPU stand-ins): same table, the
    chunks dealt to the refiners in order, a lone last chunk handled by the first refiner"""
    from sdflabel_amd.parallel import refine_sharded
    a, b = _FakeRefiner(4), _FakeRefiner(4)
    tm = {}
    table = refine_sharded([a, b], _params(19), torch.zeros(19, 3, 4, 4), [torch.zeros(2, 3)] * 19, 3, timing=tm)
    assert torch.equal(table, _expected(19, 3))
    assert len(a.calls) == 3 and len(b.calls) == 2 and tm["chunks"] == 5 and tm["refiners_in_flight"] == 2
    assert a.calls[0][0] == 0 and b.calls[0][0] == 4 and a.calls[1][0] == 8 and b.calls[1][0] == 12 and a.calls[2][0] == 16
    with pytest.raises(ValueError, match="share batch size"):
        refine_sharded([_FakeRefiner(4), _FakeRefiner(2)], _params(5), torch.zeros(1, 3, 4, 4), torch.zeros(2, 3), 1)
