"""GPU tests of the batched, sync-free path (BatchRenderer) against the drop-in modules (which are pinned to the reference goldens)
and against the golden end-to-end gradients directly."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import sdflabel_amd
from tests._util import ASSET, K_for, gold
from tests.test_gpu_parity import N, T, build_pose, images_close

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def dec():
    d, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=torch.float32)
    return d.to(DEV)


def dropin_step(dec, D, H, W, K, yaw0, trans0, lat0, weights):
    grid = sdflabel_amd.Grid3D(D, DEV)
    lat = T(np.asarray(lat0, np.float32)).requires_grad_(True)
    yaw = T(np.asarray([yaw0], np.float32)).requires_grad_(True)
    trans = T(np.asarray(trans0, np.float32)).requires_grad_(True)
    r = sdflabel_amd.Rasterer(T(K), (W, H)).to(DEV)
    lat_ = F.normalize(lat, p=2, dim=0)
    inputs = torch.cat([lat_.expand(grid.points.size(0), -1), grid.points], 1)
    sdf, _ = dec(inputs)
    pcd, _, nrm = grid.get_surface_points(sdf)
    rend, pts = r(pcd, nrm, nrm, build_pose(yaw, trans), rot="dcm", output_mask=True, output_depth=True, output_normals=True,
                  output_nocs=True)
    loss = sum((rend[k] * weights[k]).sum() for k in ("color", "mask", "depth", "normals"))
    nf = pts["xyzf"].shape[0]
    loss = loss + (pts["xyzf"] * weights["xyzf"][:nf]).sum()
    loss.backward()
    return rend, pts, pcd.shape[0], (yaw.grad, trans.grad, lat.grad)


@pytest.mark.parametrize("B,D,H,W", [(3, 16, 32, 32), (2, 21, 40, 48)])
def test_batch_matches_dropin(dec, B, D, H, W):
    rng = np.random.default_rng(B * 10 + D)
    K = K_for(H, W)
    yaws = rng.uniform(-1.0, 1.0, B).astype(np.float32)
    trans = np.stack([rng.uniform(-0.2, 0.2, B), rng.uniform(-0.15, 0.15, B), rng.uniform(2.8, 3.8, B)], 1).astype(np.float32)
    lats = rng.standard_normal((B, 3)).astype(np.float32)
    br = sdflabel_amd.BatchRenderer(dec, D, K, (W, H), B, device=DEV)
    out = br.forward(T(yaws), T(trans), T(lats))
    cap = br.cap
    w = {"color": torch.randn(B, 3, H, W, device=DEV), "mask": torch.randn(B, 1, H, W, device=DEV),
         "depth": torch.randn(B, 1, H, W, device=DEV), "normals": torch.randn(B, 3, H, W, device=DEV),
         "xyzf": torch.randn(B, cap, 3, device=DEV)}
    g_yaw, g_trans, g_lat = br.backward(g_color=w["color"], g_mask=w["mask"], g_depth=w["depth"], g_normals=w["normals"], g_xyzf=w["xyzf"])
    assert not br.overflow()
    for b in range(B):
        rend, pts, n, grads = dropin_step(dec, D, H, W, K, yaws[b], trans[b], lats[b], {k: v[b] for k, v in w.items()})
        assert int(out["n"][b]) == n and int(out["nf"][b]) == pts["xyzf"].shape[0]
        for k, kk in (("color", "color"), ("mask", "mask"), ("depth", "depth"), ("normals", "normals")):
            assert np.abs(N(out[kk][b]) - N(rend[k])).max() < 2e-5, k
        nf = pts["xyzf"].shape[0]
        assert np.abs(N(out["xyzf"][b, :nf]) - N(pts["xyzf"])).max() < 1e-5
        assert float(out["xyzf"][b, nf:].abs().max()) == 0.0
        for got, ref in ((g_yaw[b:b + 1], grads[0]), (g_trans[b], grads[1]), (g_lat[b], grads[2])):
            ref = N(ref)
            assert np.abs(N(got) - ref).max() < 2e-4 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("tag", ["a", "b"])
def test_batch_gradients_golden(dec, tag):
    """BatchRenderer against the reference's autograd directly (golden G7), crop replicated in a batch of 2."""
    z = gold("g7_grads.npz")
    D, H, W = [int(v) for v in z[tag + "_cfg"]]
    B = 2
    br = sdflabel_amd.BatchRenderer(dec, D, z[tag + "_K"], (W, H), B, device=DEV)
    yaw = T(np.repeat(z[tag + "_yaw"], B)); trans = T(np.tile(z[tag + "_trans"], (B, 1))); lat = T(np.tile(z[tag + "_latent"], (B, 1)))
    out = br.forward(yaw, trans, lat)
    for k in ("color", "mask", "depth", "normals"):
        images_close(N(out[k][1]), z[tag + "_out_" + k])
    nf = int(out["nf"][0])
    gx = torch.zeros(B, br.cap, 3, device=DEV)
    gx[:, :nf] = T(z[tag + "_Wp_xyzf"])
    # the golden functional also weights xyz / rgb / rgbf, which BatchRenderer does not expose: compare on the exposed subset by
    # subtracting nothing -- instead rebuild the same functional restricted to images + xyzf with the drop-in path
    g = br.backward(g_color=T(z[tag + "_W_color"]), g_mask=T(z[tag + "_W_mask"]), g_depth=T(z[tag + "_W_depth"]),
                    g_normals=T(z[tag + "_W_normals"]), g_xyzf=gx)
    w = {k: T(z[tag + "_W_" + k]) for k in ("color", "mask", "depth", "normals")}
    w["xyzf"] = gx[0]
    _, _, _, ref = dropin_step(dec, D, H, W, z[tag + "_K"], float(z[tag + "_yaw"][0]), z[tag + "_trans"], z[tag + "_latent"], w)
    for got, r in ((g[0][1:2], ref[0]), (g[1][1], ref[1]), (g[2][1], ref[2])):
        r = N(r)
        assert np.abs(N(got) - r).max() < 2e-4 * max(1.0, np.abs(r).max())


def test_batch_graph_replay_matches_eager(dec):
    D, H, W, B = 16, 32, 32, 2
    br = sdflabel_amd.BatchRenderer(dec, D, K_for(H, W), (W, H), B, device=DEV)
    br.set_params(T(np.array([0.5, -0.4], np.float32)), T(np.array([[0, 0, 3.4], [0.1, 0, 3.0]], np.float32)),
                  T(np.array([[0.3, -0.5, 0.8], [-0.6, 0.2, 0.1]], np.float32)))
    ones_c = torch.ones(B, 3, H, W, device=DEV)
    grads_fn = lambda o: dict(g_color=ones_c, g_xyzf=torch.ones_like(o["xyzf"]))
    br.backward(**grads_fn(br.forward()))
    ref = [t.clone() for t in (br.g_yaw, br.g_trans, br.g_latent, br.color)]
    replay = br.capture(grads_fn)
    br.g_yaw.zero_(); br.g_latent.zero_()
    replay()
    torch.cuda.synchronize()
    for a, b in zip(ref, (br.g_yaw, br.g_trans, br.g_latent, br.color)):
        assert torch.equal(a, b)
    # parameters are read from the static buffers at replay time
    br.yaw.add_(0.2)
    replay()
    torch.cuda.synchronize()
    assert not torch.equal(ref[3], br.color)


def test_batch_capacity_overflow_flag(dec):
    br = sdflabel_amd.BatchRenderer(dec, 16, K_for(16, 16), (16, 16), 1, cap=64, device=DEV)
    br.forward(T(np.array([0.6], np.float32)), T(np.array([[0, 0, 3.5]], np.float32)), T(np.array([[0.3, -0.5, 0.8]], np.float32)))
    assert br.overflow() and int(br.cnt[0]) == 167        # true band size (golden G3 'a'), only the first 64 kept


@pytest.mark.parametrize("gfile", ["g8_optimizer.npz", "g8b_optimizer_128.npz"])
@pytest.mark.parametrize("B,graph", [(1, False), (2, True)])
def test_batch_refiner_trajectory_golden(dec, B, graph, gfile):
    """f1+f2+f3: the device-resident refinement loop (HIP losses + solver step) against the trajectory of the reference's own
    Optimizer (goldens G8: 32x32, D=20; G8b: BASELINE configs[0]'s 128x128, D=40): parameters after each of 10 iterations and the
    per-iteration weighted losses."""
    z = gold(gfile)
    D, H, W = int(z["D"]), int(z["H"]), int(z["W"])
    init = z["init"]
    rf = sdflabel_amd.BatchRefiner(dec, D, z["K"], (H, W), B, lidar_cap=max(256, int(z["lidar"].shape[0])), weights={"2d": 0.3, "3d": 0.5}, device=DEV)
    rep = lambda a: np.tile(np.asarray(a, np.float32).reshape(1, -1), (B, 1))
    rf.set_crops({"yaw": rep(init[0:1]), "trans": rep(init[1:4]), "scale": rep(init[4:5]), "latent": rep(init[5:8])},
                 np.tile(z["nocs_target"][None], (B, 1, 1, 1)), [z["lidar"]] * B)
    if graph:
        rf.capture()
    traj, l2, l3 = [], [], []
    for _ in range(10):
        rf.optimize(1)
        rows, a, b = rf.results()
        traj.append(N(rows)); l2.append(N(a)); l3.append(N(b))
        assert int(rf.stepped.min()) == 1
    traj, l2, l3 = np.asarray(traj), np.asarray(l2), np.asarray(l3)
    for b in range(B):
        assert np.abs(l2[:, b] - z["loss2d_weighted"]).max() < 2e-4
        assert np.abs(l3[:, b] - z["loss3d_weighted"]).max() < 2e-4
        assert np.abs(traj[:, b] - z["traj"]).max() < 5e-4, np.abs(traj[:, b] - z["traj"]).max(axis=0)


@pytest.mark.parametrize("graph", [False, True])
def test_batch_refiner_follows_the_reference_for_the_full_60_iterations(dec, graph):
    """the reference's refinement length (config_refine.ini:15): the reference Optimizer's own 60-iteration run (golden G8c) ends at
    yaw 0.599 / z 3.512 (targets 0.6 / 3.5); the device-resident loop stays on that trajectory all the way and ends at the same pose, scale
    and latent.  Tolerance 1e-3 on every parameter at every iteration (the two runs differ by float rounding in 60 chained Adam / SGD steps)."""
    z = gold("g8c_optimizer_60it.npz")
    D, H, W = int(z["D"]), int(z["H"]), int(z["W"])
    init = z["init"]
    rf = sdflabel_amd.BatchRefiner(dec, D, z["K"], (H, W), 1, lidar_cap=256, weights={"2d": 0.3, "3d": 0.5}, device=DEV)
    rf.set_crops({"yaw": init[None, 0:1], "trans": init[None, 1:4], "scale": init[None, 4:5], "latent": init[None, 5:8]},
                 z["nocs_target"][None], [z["lidar"]])
    if graph:
        rf.capture()
    traj = []
    for _ in range(60):
        rf.optimize(1)
        traj.append(N(rf.results()[0])[0])
    traj = np.asarray(traj)
    dev = np.abs(traj - z["traj"]).max(axis=1)
    assert dev.max() < 1e-3, (int(dev.argmax()), dev.max())
    assert np.abs(traj[-1] - z["traj"][-1]).max() < 1e-3
    assert abs(traj[-1, 0] - 0.6) < 5e-3 and abs(traj[-1, 3] - 3.5) < 2e-2          # and that is the target pose


@pytest.mark.parametrize("case", ["far", "nolidar", "blank"])
def test_batch_refiner_skip_rules_and_degenerate_losses_golden(dec, case):
    """the reference Optimizer's own runs (golden G8s) of: lidar out of reach (3-D loss 0, the step uses the 2-D loss alone -- whose
    pose gradient is ~1e-9, so Adam moves yaw by 1e-5 per step, not 1e-2); no lidar at all ('Skip frame' every iteration, optimizer.py:127-129,
    parameters untouched); an all-zero target image."""
    z = gold("g8s_optimizer_skips.npz")
    D, H, W = int(z["D"]), int(z["H"]), int(z["W"])
    init = z["init"]
    rf = sdflabel_amd.BatchRefiner(dec, D, z["K"], (H, W), 1, lidar_cap=256, weights={"2d": 0.3, "3d": 0.5}, device=DEV)
    rf.set_crops({"yaw": init[None, 0:1], "trans": init[None, 1:4], "scale": init[None, 4:5], "latent": init[None, 5:8]},
                 z[case + "_target"][None], [z[case + "_lidar"]])
    for it in range(4):
        rf.optimize(1)
        got = N(rf.results()[0])[0]
        assert int(rf.stepped[0]) == 1 - int(z[case + "_skipped"][it])
        ref = z[case + "_traj"][it]
        if case == "nolidar":
            assert np.array_equal(got, init)
        else:
            moved = np.abs(ref - init).max()
            assert np.abs(got - ref).max() < max(2e-2 * moved, 2e-6), (it, got - ref, moved)


def test_losses_vs_torch_restatement(dec):
    """the HIP 2-D / 3-D losses and their gradients against the torch restatement of optimizer.py:166-237 (tests/_harness.py)."""
    from sdflabel_amd import _lib
    from tests._harness import loss_2d, loss_3d
    L = _lib.lib()
    rng = np.random.default_rng(3)
    H, W = 24, 20
    rend = torch.zeros(3, H, W, device=DEV)
    rend[:, 5:17, 4:15] = torch.rand(3, 12, 11, device=DEV)
    rend.requires_grad_(True)
    tgt = torch.rand(3, H, W, device=DEV) * (torch.rand(1, H, W, device=DEV) > 0.4)
    ref = loss_2d(rend, tgt)
    ref.backward()
    loss = torch.zeros(1, device=DEV); g = torch.zeros(1, 3, H, W, device=DEV); nv = torch.zeros(1, dtype=torch.int32, device=DEV)
    scr = torch.zeros(3 * ((W + 15) // 16) * ((H + 15) // 16), device=DEV)
    _lib.check(L.sdfr_loss_2d(_lib.ptr(rend.detach().contiguous()), _lib.ptr(tgt.contiguous()), 1, H, W, 5.0, 1.0, 1.0, _lib.ptr(loss),
                              _lib.ptr(g), _lib.ptr(nv), _lib.ptr(scr), _lib.stream_ptr()), "loss2d")
    assert abs(float(loss) - float(ref)) < 1e-5
    assert np.abs(N(g[0]) - N(rend.grad)).max() < 1e-5
    # other window sizes: a generic LDS instantiation (diam 7) and, wider than the LDS tile's halo (diam > 9), the global-memory one
    for dm in (7.0, 10.5):
        rend2 = rend.detach().clone().requires_grad_(True)
        ref2 = loss_2d(rend2, tgt, diam=dm)
        ref2.backward()
        _lib.check(L.sdfr_loss_2d(_lib.ptr(rend2.detach().contiguous()), _lib.ptr(tgt.contiguous()), 1, H, W, dm, 1.0, 1.0, _lib.ptr(loss),
                                  _lib.ptr(g), _lib.ptr(nv), _lib.ptr(scr), _lib.stream_ptr()), "loss2d")
        assert abs(float(loss) - float(ref2)) < 1e-5, dm
        assert np.abs(N(g[0]) - N(rend2.grad)).max() < 1e-5, dm
    # 3-D
    est = torch.rand(300, 3, device=DEV).requires_grad_(True)
    lidar = torch.rand(150, 3, device=DEV) * 2.0
    scale = torch.tensor([2.0], device=DEV, requires_grad=True)
    ref = loss_3d(est, lidar / scale, float(scale))
    ref.backward()
    cap = 512
    estp = torch.zeros(1, cap, 3, device=DEV); estp[0, :300] = est.detach()
    lid = torch.zeros(1, 256, 3, device=DEV); lid[0, :150] = lidar
    ec = torch.tensor([300], dtype=torch.int32, device=DEV); lc = torch.tensor([150], dtype=torch.int32, device=DEV)
    l3 = torch.zeros(1, device=DEV); ge = torch.zeros(1, cap, 3, device=DEV); gs = torch.zeros(1, device=DEV)
    npair = torch.zeros(1, dtype=torch.int32, device=DEV)
    scr3 = torch.zeros(3 * ((cap + 63) // 64), device=DEV); ge += 7.0          # the call must overwrite every row of g_est
    _lib.check(L.sdfr_loss_3d(_lib.ptr(estp), _lib.ptr(ec), cap, _lib.ptr(lid), _lib.ptr(lc), 256, _lib.ptr(scale.detach()), 0.2, 1.0, 1,
                              _lib.ptr(l3), _lib.ptr(ge), _lib.ptr(gs), _lib.ptr(npair), _lib.ptr(scr3), _lib.stream_ptr()), "loss3d")
    assert int(npair) > 10
    assert float(ge[0, 300:].abs().max()) == 0.0                     # rows beyond the count carry no gradient
    assert abs(float(l3) - float(ref)) < 1e-6
    assert np.abs(N(ge[0, :300]) - N(est.grad)).max() < 1e-6
    assert abs(float(gs) - float(scale.grad)) < 1e-5


def test_batch_of_64_is_bitwise_the_single_crop(dec):
    """BASELINE configs[2] shape (64 crops per launch): every crop of the batch equals the same crop rendered alone, bit for bit
    (tiles of the decoder kernel never mix crops; all reductions are per crop and fixed-order)."""
    D, H, W, B = 20, 64, 64, 64
    rng = np.random.default_rng(64)
    yaws = rng.uniform(-1.0, 1.0, B).astype(np.float32)
    trans = np.stack([rng.uniform(-0.2, 0.2, B), rng.uniform(-0.15, 0.15, B), rng.uniform(2.8, 3.8, B)], 1).astype(np.float32)
    lats = rng.standard_normal((B, 3)).astype(np.float32)
    K = K_for(H, W)
    big = sdflabel_amd.BatchRenderer(dec, D, K, (W, H), B, device=DEV)
    one = sdflabel_amd.BatchRenderer(dec, D, K, (W, H), 1, device=DEV)
    ob = big.forward(T(yaws), T(trans), T(lats))
    gb = big.backward(g_color=torch.ones(B, 3, H, W, device=DEV), g_xyzf=torch.ones(B, big.cap, 3, device=DEV))
    gb = [t.clone() for t in gb]
    for b in (0, 17, 63):
        o1 = one.forward(T(yaws[b:b + 1]), T(trans[b:b + 1]), T(lats[b:b + 1]))
        g1 = one.backward(g_color=torch.ones(1, 3, H, W, device=DEV), g_xyzf=torch.ones(1, one.cap, 3, device=DEV))
        for k in ("color", "mask", "depth", "normals"):
            assert torch.equal(ob[k][b], o1[k][0]), k
        assert torch.equal(gb[0][b], g1[0][0]) and torch.equal(gb[1][b], g1[1][0]) and torch.equal(gb[2][b], g1[2][0])


def test_refinement_converges_from_perturbed_pose(dec):
    """size-independent property of the whole loop: rendering the ground-truth pose and refining from a perturbed start reduces the
    pose error (encode -> perturb -> decode round trip)."""
    D, H, W, B = 30, 64, 64, 4
    K = K_for(H, W)
    gt = sdflabel_amd.BatchRenderer(dec, D, K, (W, H), 1, device=DEV)
    o = gt.forward(T(np.array([0.6], np.float32)), T(np.array([[0.0, 0.0, 3.5]], np.float32)), T(np.array([[0.3, -0.5, 0.8]], np.float32)))
    nf = int(o["nf"][0])
    lidar = N(o["xyzf"][0, :nf] * 2.0)[::2]
    rf = sdflabel_amd.BatchRefiner(dec, D, K, (H, W), B, lidar_cap=2048, device=DEV)
    rng = np.random.default_rng(5)
    yaw0 = (0.6 + rng.uniform(0.08, 0.15, B) * rng.choice([-1, 1], B)).astype(np.float32)
    t0 = (np.array([[0.0, 0.0, 3.5]]) + rng.uniform(-0.04, 0.04, (B, 3))).astype(np.float32)
    rf.set_crops({"yaw": yaw0, "trans": t0, "scale": np.full(B, 2.0, np.float32), "latent": np.tile([[0.3, -0.5, 0.8]], (B, 1))},
                 N(o["color"]).repeat(B, 0), [lidar] * B)
    rf.capture()
    rf.optimize(40)
    rows, _, _ = rf.results()
    err0 = np.abs(yaw0 - 0.6)
    err1 = np.abs(N(rows)[:, 0] - 0.6)
    assert (err1 < 0.5 * err0).all(), (err0, err1)
    assert np.abs(N(rows)[:, 1:4] - np.array([0.0, 0.0, 3.5])).max() < 0.06


def test_split_decoder_batched_path_passes_the_float32_goldens():
    """BatchRenderer / BatchRefiner on the error-compensated f16 decoder against the float32 goldens (G7 gradients, G8 trajectory)"""
    d, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision="float32_split")
    d = d.to(DEV)
    for tag in ("a", "b"):
        test_batch_gradients_golden(d, tag)
    test_batch_refiner_trajectory_golden(d, 2, True, "g8_optimizer.npz")


@pytest.mark.parametrize("gfile", ["g8_optimizer.npz", "g8b_optimizer_128.npz"])
def test_optimizer_mirror_trajectory_end_state(dec, gfile):
    """the product-side Optimizer on both reference trajectories (G8, and G8b at BASELINE configs[0]'s size)"""
    from sdflabel_amd.pipelines.optimizer import Optimizer
    z = gold(gfile)
    D, H, W = int(z["D"]), int(z["H"]), int(z["W"])
    init = z["init"]
    params = {"yaw": init[0:1].copy(), "trans": init[1:4].copy(), "scale": init[4:5].copy(), "latent": init[5:8].copy()}
    opt = Optimizer(params, DEV, {"2d": 0.3, "3d": 0.5})
    opt.optimize(10, T(z["nocs_target"]), z["lidar"], dec, sdflabel_amd.Grid3D(D, DEV), T(z["K"]), (H, W))
    got = np.concatenate([N(params[k]).reshape(-1) for k in ("yaw", "trans", "scale", "latent")])
    assert np.abs(got - z["traj"][-1]).max() < 5e-4, np.abs(got - z["traj"][-1])
    assert abs(got[0] - init[0]) > 0.05


def test_optimizer_mirror_keeps_its_solver_state_between_calls(dec):
    """the reference builds its Adam / SGD solver in Optimizer.__init__ (optimizer.py:47-52): ten optimize(1, ...) calls of one object are
    the same ten iterations as one optimize(10, ...) call -- which is how golden G8 was recorded.  Every iteration within 1e-5."""
    from sdflabel_amd.pipelines.optimizer import Optimizer
    z = gold("g8_optimizer.npz")
    D, H, W = int(z["D"]), int(z["H"]), int(z["W"])
    init = z["init"]
    params = {"yaw": init[0:1].copy(), "trans": init[1:4].copy(), "scale": init[4:5].copy(), "latent": init[5:8].copy()}
    opt = Optimizer(params, DEV, {"2d": 0.3, "3d": 0.5})
    grid = sdflabel_amd.Grid3D(D, DEV)
    for it in range(10):
        opt.optimize(1, T(z["nocs_target"]), z["lidar"], dec, grid, T(z["K"]), (H, W))
        got = np.concatenate([N(params[k]).reshape(-1) for k in ("yaw", "trans", "scale", "latent")])
        assert np.abs(got - z["traj"][it]).max() < 1e-5, (it, np.abs(got - z["traj"][it]))
    # a fresh Optimizer object starts with a fresh solver, whatever refiner it shares
    p2 = {"yaw": init[0:1].copy(), "trans": init[1:4].copy(), "scale": init[4:5].copy(), "latent": init[5:8].copy()}
    o2 = Optimizer(p2, DEV, {"2d": 0.3, "3d": 0.5})
    o2.optimize(1, T(z["nocs_target"]), z["lidar"], dec, grid, T(z["K"]), (H, W))
    assert o2._refiner is opt._refiner
    got = np.concatenate([N(p2[k]).reshape(-1) for k in ("yaw", "trans", "scale", "latent")])
    assert np.abs(got - z["traj"][0]).max() < 1e-5


@pytest.mark.parametrize("verbose", [False, True])
def test_optimizer_mirror_reaches_the_reference_optimizers_parameters(dec, verbose, capsys):
    """sdflabel_amd.pipelines.optimizer.Optimizer, called the way refine_css_demo.py:157-191 calls the reference's: after 10 iterations the
    caller's params dict holds what the reference's own Optimizer produced (golden G8); verbose=True prints its per-iteration line."""
    from sdflabel_amd.pipelines.optimizer import Optimizer
    z = gold("g8_optimizer.npz")
    D, H, W = int(z["D"]), int(z["H"]), int(z["W"])
    init = z["init"]
    params = {"yaw": init[0:1].copy(), "trans": init[1:4].copy(), "scale": init[4:5].copy(), "latent": init[5:8].copy()}
    opt = Optimizer(params, DEV, {"2d": 0.3, "3d": 0.5})
    assert all(torch.is_tensor(v) and v.requires_grad and v.dtype == torch.float32 for v in params.values())
    grid = sdflabel_amd.Grid3D(D, DEV)
    out = opt.optimize(10, T(z["nocs_target"]), z["lidar"], dec, grid, T(z["K"]), (H, W), verbose=verbose)
    assert out is params
    got = np.concatenate([N(params[k]).reshape(-1) for k in ("yaw", "trans", "scale", "latent")])
    assert np.abs(got - z["traj"][-1]).max() < 5e-4, np.abs(got - z["traj"][-1])
    if verbose:
        l = np.asarray(opt.log)
        assert np.abs(l[:, 0] - z["loss2d_weighted"]).max() < 2e-4 and np.abs(l[:, 1] - z["loss3d_weighted"]).max() < 2e-4
        assert capsys.readouterr().out.count("ITER") == 10
    # the next crop gets a new Optimizer object, as refine_css.py:203 constructs one per crop (fresh solver state), and reuses the
    # refiner (same decoder, grid, K, crop size)
    rf = opt._refiner
    params2 = {"yaw": init[0:1].copy(), "trans": init[1:4].copy(), "scale": init[4:5].copy(), "latent": init[5:8].copy()}
    opt2 = Optimizer(params2, DEV, {"2d": 0.3, "3d": 0.5})
    opt2.optimize(10, T(z["nocs_target"]), z["lidar"], dec, grid, T(z["K"]), (H, W))
    assert opt2._refiner is rf
    got2 = np.concatenate([N(params2[k]).reshape(-1) for k in ("yaw", "trans", "scale", "latent")])
    assert np.abs(got2 - z["traj"][-1]).max() < 5e-4


def test_prefilter_two_stage_evaluation_reproduces_the_exact_path(dec):
    """precision="float32_prefilter": a half-operand pass over the grid only proposes candidates; band membership, sdf at the band rows and
    the Jacobian come from exact-f32 kernels on the candidates.  The band must be the same rows as the exact path's, images and
    gradients agree to summation-order noise, and the float32 goldens (G7 gradients, G8 trajectory) pass at the float32 tolerances."""
    dp, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision="float32_prefilter")
    dp = dp.to(DEV)
    D, H, W, B = 40, 96, 96, 2
    K = K_for(H, W)
    yaw = T(np.array([0.6, -0.4], np.float32)); trans = T(np.array([[0.0, 0.0, 3.5], [0.1, -0.05, 3.2]], np.float32))
    lat = T(np.array([[0.3, -0.5, 0.8], [-0.2, 0.6, 0.4]], np.float32))
    outs = []
    for d in (dec, dp):
        br = sdflabel_amd.BatchRenderer(d, D, K, (W, H), B, device=DEV)
        assert br.prefilter == (d is dp)
        if br.prefilter:
            assert 0.0 < br.f16_error < 2e-3 and br.margin >= 4.0 * br.f16_error       # calibrated on this decoder at construction
        o = br.forward(yaw, trans, lat)
        g = br.backward(g_color=torch.ones(B, 3, H, W, device=DEV), g_mask=torch.ones(B, 1, H, W, device=DEV),
                        g_xyzf=torch.ones(B, br.cap, 3, device=DEV))
        assert not br.overflow()
        outs.append((br.cnt.clone(), br.idx.clone(), {k: o[k].clone() for k in ("color", "mask", "depth", "normals")}, [t.clone() for t in g], br))
    (c0, i0, o0, g0, b0), (c1, i1, o1, g1, b1) = outs
    assert torch.equal(c0, c1)
    for b in range(B):
        n = int(c0[b])
        assert n > 1000 and torch.equal(i0[b, :n], i1[b, :n])
        assert int(b1.ccnt[b]) > n                                           # the candidates are a strict superset of the band
        rows = i0[b, :n].long() + b * b0.G
        assert float((b0.sdf[rows] - b1.sdf[rows]).abs().max()) < 1e-6      # exact values at the band rows (other rows stay half-accurate)
        # the exact pass sums in another order than the grid kernel: a hidden unit whose pre-activation is within rounding of zero may get
        # the other ReLU mask bit, which moves that row's Jacobian at the 1e-3 level (one row in a few thousand); all others agree to rounding
        dJ = (b0.J[b, :n] - b1.J[b, :n]).abs().max(dim=1)[0]
        assert float(dJ.max()) < 2e-2 and int((dJ > 2e-5).sum()) <= max(2, n // 500)
    for k in o0:                                                             # the pixels under such a surfel inherit its normal's change
        d = (o0[k] - o1[k]).abs()
        assert float(d.max()) < 5e-3 and float((d > 1e-4).float().mean()) < 1e-3, k
    for a, bb in zip(g0, g1):
        assert float((a - bb).abs().max()) < 1e-3 * max(1.0, float(a.abs().max()))
    for tag in ("a", "b"):
        test_batch_gradients_golden(dp, tag)
    test_batch_refiner_trajectory_golden(dp, 1, True, "g8_optimizer.npz")


@pytest.mark.parametrize("precision", [torch.float32, torch.float16, "float32_split", "float32_prefilter"])
def test_empty_band_renders_nothing_and_skips_the_crop(precision):
    """no grid point inside the band (threshold ~ 0): zero surfels -> zero images, zero finite gradients, and the refinement loop's skip
    rule (optimizer.py:127-129) holds the crop's parameters still, in every decoder precision"""
    d, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=precision)
    d = d.to(DEV)
    D, H, W, B = 20, 40, 40, 2
    K = K_for(H, W)
    br = sdflabel_amd.BatchRenderer(d, D, K, (W, H), B, device=DEV, threshold=1e-12)
    if br.prefilter:
        br.margin = 0.0
    o = br.forward(T(np.array([0.3, -0.2], np.float32)), T(np.array([[0, 0, 3.5], [0.1, 0, 3.2]], np.float32)),
                   T(np.array([[0.3, -0.5, 0.8], [0.1, 0.2, 0.9]], np.float32)))
    assert int(o["n"].max()) == 0 and int(o["nf"].max()) == 0
    for k in ("color", "mask", "depth", "normals"):
        assert float(o[k].abs().max()) == 0.0
    g = br.backward(g_color=torch.ones(B, 3, H, W, device=DEV), g_mask=torch.ones(B, 1, H, W, device=DEV),
                    g_xyzf=torch.ones(B, br.cap, 3, device=DEV))
    for t in g:
        assert torch.isfinite(t).all() and float(t.abs().max()) == 0.0
    rf = sdflabel_amd.BatchRefiner(d, D, K, (H, W), B, lidar_cap=64, device=DEV)
    rf.br.thr = 1e-12
    if rf.br.prefilter:
        rf.br.margin = 0.0
    p0 = {"yaw": np.array([[0.3], [-0.2]], np.float32), "trans": np.array([[0, 0, 3.5], [0.1, 0, 3.2]], np.float32),
          "scale": np.array([[2.0], [2.0]], np.float32), "latent": np.array([[0.3, -0.5, 0.8], [0.1, 0.2, 0.9]], np.float32)}
    rf.set_crops(p0, np.random.default_rng(0).random((B, 3, H, W)).astype(np.float32), [np.random.default_rng(1).random((30, 3)).astype(np.float32)] * B)
    before = rf.params.clone()
    rf.optimize(3)
    assert int(rf.stepped.sum()) == 0 and torch.equal(rf.params, before)


def test_fused_backward_tail_is_bitwise_the_three_kernels(dec):
    """sdfr_pose_latent_backward (one launch) against sdfr_project_dcm_bwd + sdfr_surface_latent_grad + sdfr_params_backward"""
    D, H, W, B = 40, 64, 64, 3
    K = K_for(H, W)
    br = sdflabel_amd.BatchRenderer(dec, D, K, (W, H), B, device=DEV)
    rng = np.random.default_rng(77)
    br.forward(T(rng.uniform(-1, 1, B).astype(np.float32)),
               T(np.stack([rng.uniform(-0.2, 0.2, B), rng.uniform(-0.1, 0.1, B), rng.uniform(3.0, 3.8, B)], 1).astype(np.float32)),
               T(rng.standard_normal((B, 3)).astype(np.float32)))
    gc = torch.randn(B, 3, H, W, device=DEV); gm = torch.randn(B, 1, H, W, device=DEV); gn = torch.randn(B, 3, H, W, device=DEV)
    gx = torch.randn(B, br.cap, 3, device=DEV)
    res = []
    for fused in (True, False):
        br.fused_tail = fused
        g = br.backward(g_color=gc, g_mask=gm, g_normals=gn, g_xyzf=gx)
        res.append([t.clone() for t in g] + [br.g_pose.clone(), br.g_latn.clone()])
    for a, b in zip(*res):
        assert torch.equal(a, b)
    assert float(res[0][0].abs().max()) > 0 and float(res[0][2].abs().max()) > 0


def test_fused_surfel_forward_is_bitwise_the_three_launches(dec):
    """sdfr_surfels_forward (+ SDFR_PRIM_BOXES_READY) against sdfr_surface_project + sdfr_project_dcm + the splat's own box pass"""
    D, H, W, B = 40, 72, 56, 3
    K = K_for(H, W)
    rng = np.random.default_rng(78)
    args = (T(rng.uniform(-1, 1, B).astype(np.float32)),
            T(np.stack([rng.uniform(-0.2, 0.2, B), rng.uniform(-0.1, 0.1, B), rng.uniform(3.0, 3.8, B)], 1).astype(np.float32)),
            T(rng.standard_normal((B, 3)).astype(np.float32)))
    snaps = []
    for fused, binned in ((True, True), (False, True), (True, False), (False, False)):
        br = sdflabel_amd.BatchRenderer(dec, D, K, (W, H), B, device=DEV)
        br.fused_head, br.binned = fused, binned
        o = br.forward(*args)
        n = [int(v) for v in br.cnt]; nf = [int(v) for v in br.fcnt]
        per = []
        for b in range(B):
            per.append([t[b, :n[b]].clone() for t in (br.points, br.normals, br.p_cam, br.n_cam, br.attr, br.fslot, br.boxes)] +
                       [br.fidx[b, :nf[b]].clone(), br.xyzf[b, :nf[b]].clone()])
        snaps.append((n, nf, per, {k: o[k].clone() for k in ("color", "mask", "depth", "normals")}))
    assert snaps[0][0] == snaps[1][0] and snaps[0][1] == snaps[1][1] and min(snaps[0][0]) > 500
    for other in snaps[1:]:
        assert snaps[0][0] == other[0] and snaps[0][1] == other[1]
        for pa, pb in zip(snaps[0][2], other[2]):
            for a, b in zip(pa, pb):
                assert torch.equal(a, b)
        for k in snaps[0][3]:
            assert torch.equal(snaps[0][3][k], other[3][k]), k


def test_pose_only_refinement_caches_the_shape_exactly(dec):
    """optimize_latent=False (BASELINE configs[1]: "pose-only refinement"): decoder, band and Jacobian are evaluated once per set_crops();
    the trajectory is bit-identical to re-evaluating them every iteration with the latent held fixed, eager and through the HIP graph,
    and the latent does not move."""
    z = gold("g8_optimizer.npz")
    D, H, W = int(z["D"]), int(z["H"]), int(z["W"])
    init = z["init"]
    B = 2
    rep = lambda a: np.tile(np.asarray(a, np.float32).reshape(1, -1), (B, 1))
    p0 = {"yaw": rep(init[0:1]) + np.array([[0.0], [0.05]], np.float32), "trans": rep(init[1:4]), "scale": rep(init[4:5]), "latent": rep(init[5:8])}
    runs = []
    for freeze, graph in ((False, False), (True, False), (True, True)):
        rf = sdflabel_amd.BatchRefiner(dec, D, z["K"], (H, W), B, lidar_cap=256, device=DEV, optimize_latent=False)
        rf.br.freeze_shape = freeze
        rf.set_crops(p0, np.tile(z["nocs_target"][None], (B, 1, 1, 1)), [z["lidar"]] * B)
        if graph:
            rf.capture()
        rf.optimize(8)
        rows, l2, l3 = rf.results()
        runs.append((N(rows), N(l2), N(l3)))
        if freeze:                                   # the frozen step skips the latent gradient (sdfr_pose_latent_backward with J = NULL): zeros
            assert float(rf.br.g_latent.abs().max()) == 0.0 and float(rf.br.g_yaw.abs().min()) > 0.0
        if freeze:                                   # a second set of crops through the same (captured) refiner: the new latent is picked up
            p1 = dict(p0); p1["latent"] = rep([0.2, 0.6, -0.4])
            rf.set_crops(p1, np.tile(z["nocs_target"][None], (B, 1, 1, 1)), [z["lidar"]] * B)
            rf.optimize(3)
            r1 = N(rf.results()[0])
            assert np.array_equal(r1[:, 5:8], p1["latent"]) and not np.array_equal(r1[:, 0], runs[-1][0][:, 0])
    for r in runs[1:]:
        for a, b in zip(runs[0], r):
            assert np.array_equal(a, b)
    assert np.array_equal(runs[0][0][:, 5:8], p0["latent"])                      # latent untouched
    assert np.abs(runs[0][0][:, 0] - p0["yaw"][:, 0]).min() > 0.02               # the pose moved
    # and the pose trajectory differs from the joint refinement's only through the (tiny, lr 3e-5) latent updates
    rj = sdflabel_amd.BatchRefiner(dec, D, z["K"], (H, W), B, lidar_cap=256, device=DEV)
    rj.set_crops(p0, np.tile(z["nocs_target"][None], (B, 1, 1, 1)), [z["lidar"]] * B)
    rj.optimize(8)
    assert np.abs(N(rj.results()[0])[:, :5] - runs[0][0][:, :5]).max() < 5e-3


def test_band_jacobian_variants_return_identical_bits(dec):
    """the band Jacobian runs 16-row tiles for few crops per launch and 32-row tiles (32x32x2 MFMA, paired K order) from 8 crops: the
    rows of a crop must be the same bits whichever kernel produced them"""
    D, H, W = 40, 32, 32
    rng = np.random.default_rng(11)
    B = 9
    yaws = rng.uniform(-1.0, 1.0, B).astype(np.float32)
    trans = np.tile(np.array([[0.0, 0.0, 3.5]], np.float32), (B, 1))
    lats = rng.standard_normal((B, 3)).astype(np.float32)
    big = sdflabel_amd.BatchRenderer(dec, D, K_for(H, W), (W, H), B, device=DEV)
    one = sdflabel_amd.BatchRenderer(dec, D, K_for(H, W), (W, H), 1, device=DEV)
    big.forward(T(yaws), T(trans), T(lats))
    for b in (0, 4, 8):
        one.forward(T(yaws[b:b + 1]), T(trans[b:b + 1]), T(lats[b:b + 1]))
        n = int(one.cnt[0])
        assert n == int(big.cnt[b]) and n > 1000
        assert torch.equal(one.J[0, :n], big.J[b, :n]) and torch.equal(one.sdf_band[0, :n], big.sdf_band[b, :n])


def test_prefilter_fuzz_1000_latents_band_equals_exact_and_guard_stays_quiet(dec):
    """float32_prefilter over 1000 random latents (any direction and norm: the optimizer normalises, optimizer.py:96) and poses at the
    BASELINE grid (D = 40): the band must be the exact path's rows for every crop, the run-time guard must record no violation, and the
    half pass's largest deviation at the candidates must stay far below the margin."""
    dp, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision="float32_prefilter")
    dp = dp.to(DEV)
    B, D, H, W = 8, 40, 32, 32
    K = K_for(H, W)
    b0 = sdflabel_amd.BatchRenderer(dec, D, K, (W, H), B, device=DEV)
    b1 = sdflabel_amd.BatchRenderer(dp, D, K, (W, H), B, device=DEV)
    rng = np.random.default_rng(1000)
    worst, n_crops = 0.0, 0
    for it in range(125):
        lat = (rng.standard_normal((B, 3)) * rng.choice([1e-3, 0.3, 1.0, 30.0], (B, 1))).astype(np.float32)
        if it % 10 == 0:
            lat[0] = np.eye(3, dtype=np.float32)[it // 10 % 3] * (1 if it % 20 else -1)            # axis-aligned extremes
        yaw = rng.uniform(-3, 3, B).astype(np.float32)
        tr = np.stack([rng.uniform(-0.3, 0.3, B), rng.uniform(-0.2, 0.2, B), rng.uniform(2.5, 4.5, B)], 1).astype(np.float32)
        a = [T(x) for x in (yaw, tr, lat)]
        b0.forward(*a)
        b1.forward(*a)
        n0, n1 = N(b0.cnt), N(b1.cnt)
        assert np.array_equal(n0, n1), (it, n0, n1)
        for b in range(B):
            assert torch.equal(b0.idx[b, :n0[b]], b1.idx[b, :n1[b]]), (it, b)
        worst = max(worst, float(b1.max_dev.max()))
        n_crops += B
    rep = b1.prefilter_report()
    assert n_crops == 1000 and rep["violations"] == 0 and rep["hard_violations"] == 0, rep
    assert worst < 0.25 * b1.margin, (worst, b1.margin)
    b1.check_overflow()


def test_prefilter_guard_trips_grows_the_margin_and_raises(dec):
    """force the situation the guard exists for: a margin smaller than the half pass's deviation.  The guard must count the steps, grow
    the per-crop margin on the device (so that the next selection is safe again) and check_overflow()/results() must refuse the result."""
    dp, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision="float32_prefilter")
    dp = dp.to(DEV)
    B, D, H, W = 2, 40, 32, 32
    br = sdflabel_amd.BatchRenderer(dp, D, K_for(H, W), (W, H), B, device=DEV)
    a = [T(np.array([0.6, -0.4], np.float32)), T(np.array([[0.0, 0.0, 3.5], [0.1, -0.05, 3.2]], np.float32)),
         T(np.array([[0.3, -0.5, 0.8], [-0.2, 0.6, 0.4]], np.float32))]
    br.forward(*a)
    dev0 = float(br.max_dev.max())
    assert 0 < dev0 < br.margin / 4 and br.prefilter_report()["violations"] == 0
    br.margin_dev.fill_(dev0 * 0.2)                        # an (artificially) unsafe margin
    br.forward()                                           # (same parameters; passing them again would reset the guard state: new crops)
    rep = br.prefilter_report()
    assert rep["violations"] >= 1 and rep["hard_violations"] >= 1
    grown, devs = N(br.margin_dev), N(br.max_dev)
    assert (grown >= np.maximum(dev0 * 0.2, 4.0 * devs) * 0.999).all() and grown.max() > dev0 * 0.2     # grown to 4x the observed deviation
    with pytest.raises(sdflabel_amd.SdfrError, match="prefilter"):
        br.check_overflow()
    before = br.prefilter_report()["violations"]
    br.forward()                                           # with the grown margin the next step is quiet
    assert br.prefilter_report()["violations"] == before
    br.forward(*a)                                         # new parameters = new crops: counters and margins start clean
    assert br.prefilter_report()["violations"] == 0 and float(br.margin_dev.min()) == pytest.approx(br.margin)


@pytest.mark.parametrize("reuse,arith", [(False, "split"), (True, "split"), (False, "float32")])
def test_prefilter_audit_catches_a_band_row_the_half_pass_misplaced(dec, reuse, arith):
    """the guard compares the half pass with the exact values at the CANDIDATES only; a band row that the half pass
    misplaced by more than the margin is never proposed and stays invisible to it.  Plant exactly that -- the half pass's output of one true
    band row of crop 1 overwritten with 0.5 ("far from the surface") -- and the rotating audit of the non-candidate rows (1/16 of the grid
    per step; reference values from the float32-grade split kernel -- the default -- or the exact-f32 kernel) must count a hard violation for
    that crop within 16 steps; check_overflow() then refuses the result.  Crop 0 stays quiet.  Without the audit (decoder.prefilter_audit =
    False, the r03 behaviour) the row silently drops out of the band."""
    B, D, H, W = 2, 40, 32, 32
    a = [T(np.array([0.6, -0.4], np.float32)), T(np.array([[0.0, 0.0, 3.5], [0.1, -0.05, 3.2]], np.float32)),
         T(np.array([[0.3, -0.5, 0.8], [-0.2, 0.6, 0.4]], np.float32))]
    outcomes = {}
    for audit in (True, False):
        dp, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision="float32_prefilter")
        dp.prefilter_reuse, dp.prefilter_audit, dp.prefilter_audit_arith = reuse, audit, arith
        br = sdflabel_amd.BatchRenderer(dp.to(DEV), D, K_for(H, W), (W, H), B, device=DEV)
        assert not audit or br.audit_split == (arith == "split")
        out = br.forward(*a)
        n1 = int(out["n"][1])
        g = int(br.idx[1, n1 // 2])                                    # a true band row of crop 1 (exact |sdf| < 0.03)
        assert br.prefilter_report()["hard_violations"] == 0
        if audit:
            assert br.prefilter_report()["audit"]["rows_last_step"] > 0.9 * B * br.G / 16 - br.cap
        br.fault = (torch.tensor([br.G + g], device=DEV), torch.tensor([0.5], device=DEV))
        br.invalidate_shape()                                          # (candidate reuse: force a fresh half pass so that the fault enters the candidate selection)
        caught = None
        for step in range(16):
            br.forward()
            assert int(br.cnt[1]) == n1 - 1                            # the planted row is missing from the band in every step
            if int(br.violations[1, 1]) > 0:
                caught = step
                break
        outcomes[audit] = caught
        assert int(br.violations[0].sum()) == 0                        # crop 0 is not affected
        if audit:
            assert caught is not None and caught < 16
            with pytest.raises(sdflabel_amd.SdfrError, match="prefilter"):
                br.check_overflow()
            br.fault = None
            br.forward(*a)                                             # new crops: clean state, and no fault -> quiet for a whole rotation
            for _ in range(16):
                br.forward()
            assert br.prefilter_report()["hard_violations"] == 0 and int(br.cnt[1]) == n1
            assert 0 < br.prefilter_report()["audit"]["max_deviation_at_non_candidates"] < br.margin
    assert outcomes[False] is None                                     # the guard alone never sees it


def test_prefilter_candidate_reuse_skips_the_half_pass_and_changes_nothing(dec):
    """decoder.prefilter_reuse: while the normalised latent has moved less than margin / (4 lip) since the last half pass, that pass and the
    candidate selection are skipped (decided per crop on the device).  The refinement must be bit-identical to the plain two-stage mode, most
    steps must reuse, and a jump of the latent must force a fresh half pass."""
    z = gold("g8_optimizer.npz")
    D, H, W = 40, 64, 64
    K = K_for(H, W)
    init = z["init"]
    B = 3
    rep = lambda a: np.tile(np.asarray(a, np.float32).reshape(1, -1), (B, 1))
    gt = sdflabel_amd.BatchRenderer(dec, D, K, (W, H), 1, device=DEV)
    o = gt.forward(T(np.array([0.6], np.float32)), T(np.array([[0.0, 0.0, 3.5]], np.float32)), T(np.array([[0.3, -0.5, 0.8]], np.float32)))
    lidar = N(o["xyzf"][0, :int(o["nf"][0])] * 2.0)[::2].copy()
    target = np.repeat(N(o["color"]), B, 0)
    p0 = {"yaw": rep(init[0:1]) + np.array([[0.0], [0.05], [-0.05]], np.float32), "trans": rep([0.03, 0.02, 3.45]), "scale": rep([2.0]),
          "latent": rep(init[5:8]) + np.array([[0, 0, 0], [0.1, 0, 0], [0, -0.1, 0.1]], np.float32)}
    rows, reused = [], 0
    for reuse in (False, True):
        dp, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision="float32_prefilter")
        dp.prefilter_reuse = reuse
        rf = sdflabel_amd.BatchRefiner(dp.to(DEV), D, K, (H, W), B, lidar_cap=2048, device=DEV)
        assert rf.br.reuse == reuse and rf.br.lipschitz > 0
        rf.set_crops(p0, target, [lidar] * B)
        for it in range(30):
            rf.iteration()
            if reuse:
                reused += int(rf.br.reuse_flag.sum())
        rows.append(N(rf.results()[0]))
        if reuse:
            assert rf.br.prefilter_report()["violations"] == 0
            # a jump of the latent: the plan must order a fresh half pass for that crop only
            with torch.no_grad():
                rf.latent[1] += torch.tensor([0.5, -0.4, 0.3], device=DEV)
            rf.iteration()
            assert N(rf.br.reuse_flag).tolist() == [1, 0, 1] or N(rf.br.reuse_flag)[1] == 0
    assert np.array_equal(rows[0], rows[1])
    assert reused >= 0.8 * 29 * B, reused              # (the first step of a crop and every 17th run the half pass)
    # the plan is a device-side decision: the captured HIP graph replays it
    dp, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision="float32_prefilter")
    dp.prefilter_reuse = True
    rg = sdflabel_amd.BatchRefiner(dp.to(DEV), D, K, (H, W), B, lidar_cap=2048, device=DEV)
    rg.set_crops(p0, target, [lidar] * B)
    rg.capture()
    rg.optimize(30)
    assert np.array_equal(N(rg.results()[0]), rows[0])


@pytest.mark.parametrize("kw", [dict(latent_in=[2, 4], xyz_in_all=False), dict(latent_in=[3], xyz_in_all=True), dict(latent_in=(), xyz_in_all=False)])
def test_band_jacobian_variants_on_ragged_512_wide_decoders(kw):
    """16-row and 32-row band-Jacobian kernels on 512-wide decoders whose layers are NOT all 512 wide (300 / 400 features: partly filled
    feature tiles take the thin-layer product paths) and with other injection patterns: identical bits between the two kernels, and both
    equal to the oracle's input Jacobian."""
    from oracle import sdf_oracle as O
    torch.manual_seed(11)
    dims = [300, 512, 512, 400, 512, 512]
    d = sdflabel_amd.Decoder(3, dims=dims, norm_layers=(), weight_norm=False, **kw)
    with torch.no_grad():
        for p in d.parameters():
            p.mul_(1.3)
    d = d.to(DEV).eval()
    D, B = 10, 9
    G = D ** 3
    rng = np.random.default_rng(4)
    lats = rng.standard_normal((B, 3)).astype(np.float32)
    yaws = np.zeros(B, np.float32); trans = np.tile(np.array([[0.0, 0.0, 3.5]], np.float32), (B, 1))
    big = sdflabel_amd.BatchRenderer(d, D, K_for(16, 16), (16, 16), B, cap=G, threshold=1e9, device=DEV)       # every grid row is a band row
    one = sdflabel_amd.BatchRenderer(d, D, K_for(16, 16), (16, 16), 1, cap=G, threshold=1e9, device=DEV)
    big.forward(T(yaws), T(trans), T(lats))
    assert int(big.cnt.min()) == G
    layers = [(W, b, None) for W, b in d.effective_layers()]
    spec = dict(dims=dims, latent_in=list(kw["latent_in"]), xyz_in_all=kw["xyz_in_all"])
    for b in (0, 8):
        one.forward(T(yaws[b:b + 1]), T(trans[b:b + 1]), T(lats[b:b + 1]))
        assert torch.equal(one.J[0], big.J[b]) and torch.equal(one.sdf_band[0], big.sdf_band[b])
        inp = N(big.inputs.view(B, G, 6)[b])
        ref, cache = O.decoder_forward(layers, spec, inp, want_cache=True)
        Jref = O.decoder_backward_inputs(layers, spec, inp, cache, np.ones_like(ref))
        assert np.abs(N(big.sdf.view(B, G)[b]) - ref[:, 0]).max() < 1e-5
        # (27 M hidden units per crop: a few pre-activations sit within float rounding of 0 and take the other ReLU side in the oracle's
        # summation order -- isolated elements off by ~1e-3 of a weight product; everything else agrees to rounding)
        err = np.abs(N(big.J[b]) - Jref)
        assert np.quantile(err, 0.999) < 5e-6 and err.max() < 5e-3 and (err > 5e-5).sum() <= 40, (np.quantile(err, 0.999), err.max(), (err > 5e-5).sum())


# ---- r03: ADVICE r02 ------------------------------------------------------------------------------------------------------------------

def _refine_problem(dec, D, H, W, B):
    from sdflabel_amd.fixtures import crop_params, synthetic_targets
    K = K_for(H, W)
    nocs1, lidar = synthetic_targets(dec, D, K, H, W, DEV)
    return K, crop_params(list(range(B))), nocs1.expand(B, 3, H, W), lidar


def test_prefilter_hard_violation_does_not_poison_later_crops(dec):
    """a hard guard violation on one crop must be reported for THAT refinement only: set_crops() resets the per-crop guard state (counters,
    deviation, margins), and after check_overflow() has raised once the refiner stays usable (refiners are cached and shared)"""
    D, H, W, B = 40, 64, 64, 2
    dp, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision="float32_prefilter")
    K, p0, target, lidar = _refine_problem(dec, D, H, W, B)
    rf = sdflabel_amd.BatchRefiner(dp.to(DEV), D, K, (H, W), B, lidar_cap=2048, device=DEV)
    margin0 = float(rf.br.margin)
    rf.set_crops(p0, target, [lidar] * B)
    rf.br.margin_dev.fill_(1e-6)                           # an (artificially) unsafe margin: the next step trips the guard
    rf.iteration()
    with pytest.raises(sdflabel_amd.SdfrError, match="prefilter"):
        rf.results()
    rows_after_raise = rf.results()[0]                      # reported once; the object is usable again
    assert bool(torch.isfinite(rows_after_raise).all())
    rf.set_crops(p0, target, [lidar] * B)                  # new crops: clean guard state, calibrated margin
    assert int(rf.br.violations.sum()) == 0 and float(rf.br.margin_dev.min()) == pytest.approx(margin0) and float(rf.br.margin_dev.max()) == pytest.approx(margin0)
    rf.capture()                                           # the capture's warm-up iteration must not feed the counters either
    assert int(rf.br.violations.sum()) == 0
    rf.optimize(5)
    rows, _, _ = rf.results()
    assert rf.br.prefilter_report()["hard_violations"] == 0
    # and the result is the exact path's
    de, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=torch.float32)
    re = sdflabel_amd.BatchRefiner(de.to(DEV), D, K, (H, W), B, lidar_cap=2048, device=DEV)
    re.set_crops(p0, target, [lidar] * B)
    re.optimize(5)
    assert np.abs(N(rows) - N(re.results()[0])).max() < 1e-5


def test_batch_renderer_capture_in_pose_only_mode_re_evaluates_a_new_shape(dec):
    """freeze_shape: the captured graph holds the pose-only launches; after set_params() with another latent the replay callable must run the
    decoder stages once (eagerly) instead of replaying stale surfels"""
    D, H, W = 20, 48, 48
    br = sdflabel_amd.BatchRenderer(dec, D, K_for(H, W), (W, H), 1, device=DEV)
    br.freeze_shape = True
    yaw, trans = T(np.array([0.6], np.float32)), T(np.array([[0.0, 0.0, 3.5]], np.float32))
    la, lb = T(np.array([[0.3, -0.5, 0.8]], np.float32)), T(np.array([[-0.6, 0.2, 0.1]], np.float32))
    ones = torch.ones(1, 3, H, W, device=DEV)
    grads_fn = lambda o: dict(g_color=ones, g_xyzf=torch.ones_like(o["xyzf"]))
    br.set_params(yaw, trans, la)
    replay = br.capture(grads_fn)
    ref = sdflabel_amd.BatchRenderer(dec, D, K_for(H, W), (W, H), 1, device=DEV)
    for lat in (lb, la, lb):
        br.set_params(yaw, trans, lat)
        replay()                                            # first call after a new latent: eager full step
        br.yaw.add_(0.05)
        replay()                                            # pose moved, shape unchanged: graph replay
        torch.cuda.synchronize()
        o = ref.forward(br.yaw.clone(), trans, lat)
        g = ref.backward(**grads_fn(o))
        assert torch.equal(o["color"], br.color) and int(o["n"][0]) == int(br.cnt[0])
        assert torch.equal(g[0], br.g_yaw) and torch.equal(g[1], br.g_trans)


def test_prefilter_reuse_when_a_tile_of_the_half_pass_spans_two_crops(dec):
    """D = 20: 8000 grid rows per crop are no multiple of the half pass's 128-row tile, so tiles span a crop that reuses its candidates and
    one that does not; the flagged crop's (exact, patched) values must survive and the refinement stay bit-identical to the plain two-stage
    mode (the skip test used to look at the tile's first and last rows only)"""
    D, H, W, B = 20, 48, 48, 5
    K, p0, target, lidar = _refine_problem(dec, D, H, W, B)
    rows, mixed = [], 0
    for reuse in (False, True):
        dp, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision="float32_prefilter")
        dp.prefilter_reuse = reuse
        rf = sdflabel_amd.BatchRefiner(dp.to(DEV), D, K, (H, W), B, lidar_cap=2048, device=DEV)
        rf.set_crops(p0, target, [lidar] * B)
        for it in range(12):
            if reuse and it in (4, 9):
                with torch.no_grad():                      # crops 1 and 3 need a fresh half pass, their neighbours do not
                    rf.latent[1] += torch.tensor([0.4, -0.3, 0.2], device=DEV)
                    rf.latent[3] -= torch.tensor([0.2, 0.3, -0.4], device=DEV)
            elif it in (4, 9):
                with torch.no_grad():
                    rf.latent[1] += torch.tensor([0.4, -0.3, 0.2], device=DEV)
                    rf.latent[3] -= torch.tensor([0.2, 0.3, -0.4], device=DEV)
            rf.iteration()
            if reuse:
                flags = N(rf.br.reuse_flag).tolist()
                mixed += 0 < sum(flags) < B
                if it in (4, 9):
                    # (r05: with the proven Lipschitz bound a neighbour may be due for a full pass of its own at the same step)
                    assert flags[1] == 0 and flags[3] == 0, flags
        rows.append(N(rf.results()[0]))
        assert rf.br.prefilter_report()["hard_violations"] == 0
    assert mixed >= 2                                          # steps in which reusing and non-reusing crops shared tiles of the half pass
    assert np.array_equal(rows[0], rows[1])


# ---- r03: the sharded refinement flow of BASELINE configs[3], through the function bench.py calls ------------------------------------------

def test_config3_sharded_flow_1024_crops_world_1():
    """1024 synthetic crops of 256x256 rays through sdflabel_amd.parallel.refine_sharded at world = 1 (chunks of 64, HIP-graph replay) -- the
    function bench.py's refine_sharded sections call on every rank; float16 decoder (the reference's shipped precision) and 4 iterations to
    keep the test short.  Rows of sampled crops equal, bit for bit, the same crops refined alone (B = 1): batch- and chunk-independent."""
    from sdflabel_amd.fixtures import crop_params, synthetic_targets
    from sdflabel_amd.parallel import refine_sharded
    D, H, W, TOTAL, CHUNK, ITERS = 40, 256, 256, 1024, 64, 4
    d32, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=torch.float32)
    d16, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=torch.float16)
    d32, d16 = d32.to(DEV), d16.to(DEV)
    K = K_for(H, W)
    nocs1, lidar = synthetic_targets(d32, D, K, H, W, DEV)
    rf = sdflabel_amd.BatchRefiner(d16, D, K, (H, W), CHUNK, lidar_cap=4096, device=DEV)
    rf.set_crops(crop_params(list(range(CHUNK))), nocs1.expand(CHUNK, 3, H, W), [lidar] * CHUNK)
    rf.capture()
    params = crop_params(list(range(TOTAL)))
    table = refine_sharded(rf, params, nocs1, lidar, ITERS, rank=0, world=1)
    assert tuple(table.shape) == (TOTAL, 10) and bool(torch.isfinite(table).all())          # yaw, trans, scale, latent + the two weighted losses
    moved = (table[:, 0] - T(params["yaw"])).abs()
    assert bool((moved > 0).all()) and float(moved.mean()) > 5e-3                # every crop was refined
    one = sdflabel_amd.BatchRefiner(d16, D, K, (H, W), 1, lidar_cap=4096, device=DEV)
    for i in (0, 63, 64, 517, 1023):
        one.set_crops(crop_params([i]), nocs1, [lidar])
        one.optimize(ITERS)
        r1, l2, l3 = one.results()
        assert torch.equal(r1[0], table[i, :8]) and float(l2[0]) == float(table[i, 8]) and float(l3[0]) == float(table[i, 9]), i


@pytest.mark.parametrize("precision", [torch.float32, torch.float16])
@pytest.mark.parametrize("kw", [dict(latent_in=[2, 4], xyz_in_all=False), dict(latent_in=[3], xyz_in_all=True)])
def test_band_jacobian_pool_on_other_injection_patterns(kw, precision):
    """r06: from 12 crops per launch the mask-fed band Jacobian runs as a POOL of workgroups over the live band tiles (float32: two workgroups per
    CU; float16: one, masks in LDS, J rows assembled in LDS).  On 512-wide decoders with several latent_in layers / xyz_in_all and layers that are
    not 512 wide, a crop's Jacobian and band values must have the bits of the one-crop launch (other tile geometry, no pool), and the float32
    one must equal the oracle's input Jacobian"""
    from oracle import sdf_oracle as O
    torch.manual_seed(11)
    dims = [300, 512, 512, 400, 512, 512]
    d = sdflabel_amd.Decoder(3, dims=dims, norm_layers=(), weight_norm=False, **kw)
    with torch.no_grad():
        for p in d.parameters():
            p.mul_(1.3)
    d = d.to(DEV).eval()
    d.mlp_precision = precision
    D, B = 10, 13
    G = D ** 3
    rng = np.random.default_rng(4)
    lats = rng.standard_normal((B, 3)).astype(np.float32)
    yaws = np.zeros(B, np.float32); trans = np.tile(np.array([[0.0, 0.0, 3.5]], np.float32), (B, 1))
    big = sdflabel_amd.BatchRenderer(d, D, K_for(16, 16), (16, 16), B, cap=G, threshold=1e9, device=DEV)       # every grid row is a band row
    one = sdflabel_amd.BatchRenderer(d, D, K_for(16, 16), (16, 16), 1, cap=G, threshold=1e9, device=DEV)
    big.forward(T(yaws), T(trans), T(lats))
    assert int(big.cnt.min()) == G
    for b in (0, 5, 12):
        one.forward(T(yaws[b:b + 1]), T(trans[b:b + 1]), T(lats[b:b + 1]))
        assert torch.equal(one.J[0], big.J[b]) and torch.equal(one.sdf_band[0], big.sdf_band[b]), b
    if precision == torch.float32:
        layers = [(W, b_, None) for W, b_ in d.effective_layers()]
        spec = dict(dims=dims, latent_in=list(kw["latent_in"]), xyz_in_all=kw["xyz_in_all"])
        inp = N(big.inputs.view(B, G, 6)[12])
        ref, cache = O.decoder_forward(layers, spec, inp, want_cache=True)
        Jref = O.decoder_backward_inputs(layers, spec, inp, cache, np.ones_like(ref))
        err = np.abs(N(big.J[12]) - Jref)
        assert np.quantile(err, 0.999) < 5e-6 and err.max() < 5e-3 and (err > 5e-5).sum() <= 40
