"""r06: the fused launches of a refinement iteration (sdfr_params_plan, sdfr_band_select_ex, sdfr_mlp_forward_candidates,
sdfr_candidate_band, sdfr_losses_fused, sdfr_splat_backward_x, sdfr_pose_latent_solver: 12 launches where r05 had 21) must return the bits of
the launch sequence they replace; the frame-level Optimizer.optimize_many must return the bits of one Optimizer per annotation
(pipelines/refine_css.py:94,203-223); truncation flags must be sticky (the reference has no capacity: grid.py:64-66)."""
import numpy as np
import pytest
import torch

import sdflabel_amd
from sdflabel_amd import _lib
from tests._util import ASSET, K_for
from tests.test_gpu_parity import N, T

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _dec(precision, reuse, fused):
    d, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=precision)
    d.candidate_reuse = reuse
    d.fused_launches = fused
    return d.to(DEV)


def _problem(D, H, W, B):
    from sdflabel_amd.fixtures import crop_params, synthetic_targets
    d32, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=torch.float32)
    K = K_for(H, W)
    nocs1, lidar = synthetic_targets(d32.to(DEV), D, K, H, W, DEV)
    return K, crop_params(list(range(B))), nocs1.expand(B, 3, H, W), lidar


def _same_iteration(a, b, counters=True):
    """two BatchRefiners after the same number of iterations: everything an iteration produces, bit for bit"""
    A, Bq = a.br, b.br
    assert torch.equal(A.cnt, Bq.cnt) and torch.equal(A.fcnt, Bq.fcnt)
    live = torch.arange(A.cap, device=DEV).view(1, -1) < A.cnt.view(-1, 1)
    assert torch.equal(A.idx[live], Bq.idx[live]), "band index lists differ"
    assert torch.equal(A.sdf_band[live], Bq.sdf_band[live]) and torch.equal(A.J[live], Bq.J[live])
    for name in ("color", "mask", "depth", "nimg", "xyzf", "points", "normals", "pose", "inputs"):
        assert torch.equal(getattr(A, name), getattr(Bq, name)), name
    for name in ("loss2d", "loss3d", "total", "nvalid", "npairs", "stepped", "grads", "params", "adam_m", "adam_v", "adam_t"):
        x, y = getattr(a, name), getattr(b, name)
        assert torch.equal(x, y) or (torch.isnan(x) == torch.isnan(y)).all() and torch.equal(torch.nan_to_num(x), torch.nan_to_num(y)), name
    if A.creuse:
        assert torch.equal(A.reuse_flag, Bq.reuse_flag) and torch.equal(A.age, Bq.age)
        assert not counters or torch.equal(A.n_full, Bq.n_full)        # (capture()'s warm-up iteration counts one more full pass)
        assert torch.equal(A.ccnt, Bq.ccnt) and torch.equal(A.lat_ref, Bq.lat_ref)


@pytest.mark.parametrize("precision,reuse", [(torch.float16, True), (torch.float32, True), (torch.float16, False), (torch.float32, False)])
@pytest.mark.parametrize("B,H,W,ragged", [(1, 32, 32, True), (3, 64, 48, False), (9, 40, 56, True)])
def test_fused_launches_return_the_bits_of_the_sequence_they_replace(B, H, W, ragged, precision, reuse):
    D, iters = 40, 14
    K, p0, target, lidar = _problem(D, H, W, B)
    kw = dict(max_pixels=4096, max_side=128) if ragged else {}
    old = sdflabel_amd.BatchRefiner(_dec(precision, reuse, False), D, K, (H, W), B, lidar_cap=4096, device=DEV, **kw)
    new = sdflabel_amd.BatchRefiner(_dec(precision, reuse, True), D, K, (H, W), B, lidar_cap=4096, device=DEV, **kw)
    assert new.fused and new.br.fused and not old.fused and not old.br.fused
    tg = [target[b] for b in range(B)] if ragged else target
    for rf in (old, new):
        rf.set_crops(p0, tg, [lidar] * B)
    for it in range(iters):
        old.iteration(); new.iteration()
        _same_iteration(old, new)
        # the un-normalised gradients times their factors are the arrays the r05 finalize passes stored
        assert torch.equal(new.g_color * new.kscale[:, 0].view(-1, 1, 1) if ragged else new.g_color * new.kscale[:, 0].view(-1, 1, 1, 1), old.g_color)
        assert torch.equal(new.g_xyzf * new.kscale[:, 1].view(-1, 1, 1), old.g_xyzf)
        if it == 6:                 # a latent jump in the middle: the next step orders a full pass for that crop
            with torch.no_grad():
                for rf in (old, new):
                    rf.latent[B - 1] += torch.tensor([0.4, -0.3, 0.2], device=DEV)
    # ... and the captured graph of the fused iteration replays to the same state as the eager one
    new.set_crops(p0, tg, [lidar] * B); old.set_crops(p0, tg, [lidar] * B)
    new.capture()
    new.optimize(10)
    for _ in range(10):
        old.iteration()
    _same_iteration(old, new, counters=False)
    old.check_overflow(); new.check_overflow()


def test_optimize_many_returns_the_bits_of_one_optimizer_per_annotation():
    """pipelines/refine_css.py:94,203-223 builds one Optimizer per annotation; Optimizer.optimize_many refines a frame's annotations together
    (ragged extents: own crop size and intrinsics each).  KITTI-like crops at the reference's shipped rendering_area 32 (config_refine.ini:12),
    float16 decoder (:19), 60 iterations (:15); frames of 3 (padded to a batch of 4), 5 and 8 annotations."""
    from sdflabel_amd.fixtures import kitti_like_problems
    from sdflabel_amd.pipelines import optimizer as OP
    D, n, iters = 40, 16, 60
    d32, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=torch.float32)
    d16, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=torch.float16)
    d16 = d16.to(DEV)
    shapes, Ks, targets, lidars, starts = kitti_like_problems(d32.to(DEV), D, 32, n, DEV)
    grid = sdflabel_amd.Grid3D(D, DEV)
    W8 = {"2d": 0.3, "3d": 0.5}
    OP.clear_refiner_cache()
    single = []
    for b in range(n):
        opt = OP.Optimizer({k: v.copy() for k, v in starts[b].items()}, DEV, W8)
        single.append(opt.optimize(iters, targets[b], lidars[b], d16, grid, torch.from_numpy(Ks[b]), list(shapes[b])))
    many, c0 = [], 0
    for fsz in (3, 5, 8):
        frame = [({k: v.copy() for k, v in starts[b].items()}, targets[b], lidars[b], Ks[b], shapes[b]) for b in range(c0, c0 + fsz)]
        res = OP.Optimizer.optimize_many(frame, iters, d16, grid, DEV, W8)
        assert all(res[i] is frame[i][0] for i in range(fsz))            # the callers' params dicts, refined in place
        many += res
        c0 += fsz
    for b in range(n):
        for k in ("yaw", "trans", "scale", "latent"):
            assert many[b][k].requires_grad and many[b][k].dtype == torch.float32
            assert torch.equal(many[b][k], single[b][k]), (b, k)
    moved = np.mean([abs(float(single[b]["yaw"][0]) - 0.6) for b in range(n)]) < np.mean([abs(float(starts[b]["yaw"][0]) - 0.6) for b in range(n)])
    assert moved
    OP.clear_refiner_cache()


@pytest.mark.parametrize("reuse", [True, False])
def test_truncation_flags_are_sticky_across_graph_replays(reuse):
    """VERDICT r05 missing 4: a band that exceeds `cap` during iterations 5-12 of a graph-replayed refinement and fits again at the end must not
    pass results() -- cnt[b] is overwritten by every iteration, the flag is not.  Planted by a latent that inflates the band for a while."""
    D, H, W, B = 40, 32, 32, 2
    K, p0, target, lidar = _problem(D, H, W, B)
    probe = sdflabel_amd.BatchRefiner(_dec(torch.float16, reuse, True), D, K, (H, W), B, lidar_cap=4096, device=DEV)
    sizes = []
    lat0 = np.asarray(p0["latent"], np.float32)
    cands = [lat0, -lat0, lat0[:, ::-1].copy(), lat0 * np.float32(0.2), np.abs(lat0)]
    for lat in cands:                                         # band (and candidate) counts of a few latents: the decoder's shapes differ in size
        q = dict(p0); q["latent"] = lat
        probe.set_crops(q, target, [lidar] * B)
        probe.iteration()
        sizes.append((int(probe.br.cnt.max()), int(probe.br.ccnt.max()) if reuse else int(probe.br.cnt.max())))
    i_small = int(np.argmin([c for _, c in sizes]))
    i_big = int(np.argmax([n_ for n_, _ in sizes]))
    need, top = max(sizes[i_small]), sizes[i_big][0]
    if top - need < 8:
        pytest.skip("no pair of probe latents plants an overflow (band / candidate counts %s)" % (sizes,))
    cap = (need + top) // 2
    small_p, big_p = dict(p0), dict(p0)
    small_p["latent"], big_p["latent"] = cands[i_small], cands[i_big]
    rf = sdflabel_amd.BatchRefiner(_dec(torch.float16, reuse, True), D, K, (H, W), B, lidar_cap=4096, device=DEV, cap=cap)
    rf.set_crops(small_p, target, [lidar] * B)
    rf.capture()
    rf.optimize(5)
    rf.check_overflow()                                       # fits so far
    with torch.no_grad():
        keep = rf.latent.clone()
        rf.latent.copy_(T(np.asarray(big_p["latent"], np.float32)))
    rf.optimize(2)                                            # iterations 6-7: the band exceeds cap
    def last_counts_overflow():                               # what r05's check looked at: the LAST iteration's counts
        return bool((rf.br.cnt > cap).any()) or (reuse and bool((rf.br.ccnt > rf.br.cstride).any()))
    assert last_counts_overflow()
    with torch.no_grad():
        rf.latent.copy_(keep)
    rf.optimize(13)                                           # ... and fits again by iteration 20
    assert not last_counts_overflow()                         # the last iteration's counts alone look fine ...
    with pytest.raises(sdflabel_amd.SdfrError, match="capacity"):      # ... the sticky flags do not
        rf.results()
    rf.check_overflow()                                       # reported once
    rf.set_crops(small_p, target, [lidar] * B)                # new crops start clean
    rf.optimize(3)
    rf.results()


def test_float16_jacobian_pool_with_empty_short_and_ragged_crops_gives_the_one_crop_bits():
    """C ABI: sdfr_mlp_jacobian (mask-fed, half operands) at 16 crops per launch runs as a POOL of workgroups over the crops' LIVE band tiles (r06;
    from 12 crops).  Per-crop counts of 0, 1, 63, 64, 65, a full band, ... must map every tile to its crop: each live row gets the bits it has in the launch with the crops' full
    bands (which the B = 64 vs B = 1 refinement tests tie to the one-crop geometry), rows beyond a crop's count stay untouched, empty crops cost nothing"""
    B, D = 16, 40
    d = _dec(torch.float16, False, True)
    br = sdflabel_amd.BatchRenderer(d, D, K_for(32, 32), (32, 32), B, device=DEV)
    g = torch.Generator().manual_seed(3)
    lat = torch.tensor([[0.3, -0.5, 0.8]]) + 0.2 * (torch.rand(B, 3, generator=g) - 0.5)
    br.forward(torch.full((B,), 0.7, device=DEV), torch.tensor([[0.05, 0.02, 3.3]], device=DEV).expand(B, 3).contiguous(), lat.to(DEV))
    full = br.cnt.clone()
    assert int(full.min()) > 500
    want = br.J.clone()
    L, P = _lib.lib(), _lib.ptr
    counts = full.clone()
    for b, c in enumerate([0, 1, 63, 64, 65, None, 0, 127, None, 129, 0, 0, None, 640, 2, None]):
        if c is not None:
            counts[b] = c
    J = torch.full_like(br.J, -7.0)
    sb = torch.full_like(br.sdf_band, -7.0)
    _lib.check(L.sdfr_mlp_jacobian(br.handle.h, P(br.inputs), br.G, B, P(br.idx), br.cap, P(counts), P(J), P(sb), P(br.sdf), P(br.mask_ws), 2,
                                   _lib.stream_ptr()), "sdfr_mlp_jacobian")
    torch.cuda.synchronize()
    live = torch.arange(br.cap, device=DEV).view(1, -1) < counts.view(-1, 1)
    assert torch.equal(J[live], want[live]) and torch.equal(sb[live], br.sdf_band[live])
    assert bool((J[~live] == -7.0).all()) and bool((sb[~live] == -7.0).all())


@pytest.mark.parametrize("precision", [torch.float16, torch.float32])
def test_two_chunks_in_flight_give_the_bits_of_one_refiner(precision):
    """r06: sdflabel_amd.parallel.refine_sharded with a LIST of refiners refines that many chunks at the same time, each refiner's iterations replayed
    on a stream of its own (the decoder passes of one chunk beside the splat / loss kernels of the other).  The chunks are independent: the
    gathered table must equal the one-refiner table bit for bit -- 22 crops in chunks of 4 (a lone, padded last chunk), HIP-graph replay"""
    from sdflabel_amd.parallel import refine_sharded
    from sdflabel_amd.fixtures import crop_params
    D, H, W, B, n = 40, 48, 48, 4, 22
    K, _, target, lidar = _problem(D, H, W, 1)
    d = _dec(precision, True, True)

    def refiner():
        rf = sdflabel_amd.BatchRefiner(d, D, K, (H, W), B, lidar_cap=4096, device=DEV)
        rf.set_crops(crop_params(list(range(B))), target.expand(B, 3, H, W), [lidar] * B)
        rf.capture()
        return rf

    one = refine_sharded(refiner(), crop_params(list(range(n))), target, lidar, 12)
    tm = {}
    two = refine_sharded([refiner(), refiner()], crop_params(list(range(n))), target, lidar, 12, timing=tm)
    assert tuple(one.shape) == (n, 10) and torch.equal(one, two) and tm["chunks"] == 6 and tm["refiners_in_flight"] == 2
