"""r05: candidate reuse of the float16 decoder mode -- the reference's shipped precision (configs/config_refine.ini:19) --
and of the exact-float32 mode (the parity path) must change NOTHING: band index lists, decoder values, Jacobians and every image bit-identical
to the full-grid evaluation with the same kernel, iteration by iteration, while most iterations evaluate the candidate rows only."""
import numpy as np
import pytest
import torch

import sdflabel_amd
from sdflabel_amd import _lib
from tests._util import ASSET, K_for
from tests.test_gpu_parity import N, T

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _dec16(reuse, precision=torch.float16, **attrs):
    d, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=precision)
    d.candidate_reuse = reuse
    for k, v in attrs.items():
        setattr(d, k, v)
    return d.to(DEV)


def _problem(D, H, W, B):
    from sdflabel_amd.fixtures import crop_params, synthetic_targets
    d32, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=torch.float32)
    K = K_for(H, W)
    nocs1, lidar = synthetic_targets(d32.to(DEV), D, K, H, W, DEV)
    return K, crop_params(list(range(B))), nocs1.expand(B, 3, H, W), lidar


def _same_step(a, b):
    """every array the rest of the step consumes, bit for bit (ragged arrays up to each crop's count)"""
    assert torch.equal(a.cnt, b.cnt) and torch.equal(a.fcnt, b.fcnt)
    live = torch.arange(a.cap, device=DEV).view(1, -1) < a.cnt.view(-1, 1)
    assert torch.equal(a.idx[live], b.idx[live]), "band index lists differ"
    assert torch.equal(a.sdf_band[live], b.sdf_band[live]) and torch.equal(a.J[live], b.J[live])
    for name in ("color", "mask", "depth", "nimg", "xyzf", "points", "normals"):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
    for name in ("g_yaw", "g_trans", "g_latent"):
        assert torch.equal(getattr(a, name), getattr(b, name)), name


@pytest.mark.parametrize("precision", [torch.float16, torch.float32])
@pytest.mark.parametrize("B,H,W,iters", [(64, 256, 256, 60), (3, 64, 48, 40)])
def test_candidate_reuse_is_bit_identical_to_the_full_grid_evaluation(B, H, W, iters, precision):
    D = 40
    K, p0, target, lidar = _problem(D, H, W, B)
    plain = sdflabel_amd.BatchRefiner(_dec16(False, precision), D, K, (H, W), B, lidar_cap=4096, device=DEV)
    reuse = sdflabel_amd.BatchRefiner(_dec16(True, precision), D, K, (H, W), B, lidar_cap=4096, device=DEV)
    assert reuse.br.creuse and not plain.br.creuse and reuse.br.lipschitz > 100.0        # the proven bound, not the sampled constant
    assert reuse.br.margin >= 4.0 * reuse.br.f16_error > 0
    plain.set_crops(p0, target, [lidar] * B)
    reuse.set_crops(p0, target, [lidar] * B)
    reused = 0
    for it in range(iters):
        plain.iteration()
        reuse.iteration()
        _same_step(plain.br, reuse.br)
        flags = int(reuse.br.reuse_flag.sum())
        assert it > 0 or flags == 0                                      # a crop's first step is a full pass
        reused += flags
    assert torch.equal(plain.results()[0], reuse.results()[0])
    rep = reuse.br.prefilter_report()
    assert rep["hard_violations"] == 0 and rep["violations"] == 0
    # the audit's values at rows OUTSIDE the candidates, on steps with a full pass, are compared with that pass's: the same kernel arithmetic
    # in another launch shape -> exactly equal (float32 at a few crops: the audit's 16-row tiles sum in another order -- rounding noise)
    if precision == torch.float16:
        assert rep["audit"]["max_deviation_at_non_candidates"] == 0.0
    else:                                                    # (float32: the audit's values come from the float32-grade split kernel, the
        assert reuse.br.select_half                          #  values they are compared with from the HALF selection pass over the grid)
        assert 0 < rep["audit"]["max_deviation_at_non_candidates"] < reuse.br.margin / 4
        assert reuse.br.select_error < 1e-3
    assert reuse.br.f16_error < (1e-3 if precision == torch.float16 else 2e-6)          # the mode's kernel against exact arithmetic (budgeted)
    # most steps evaluate the candidates alone: the first step of a crop, every (max_reuse + 1)-th and the steps after the latent has moved
    # by margin / (4 lip) run the full grid
    assert reused >= 0.75 * (iters - 1) * B, (reused, iters, B)
    # ... and the candidates are a small part of the grid
    assert 0 < int(reuse.br.ccnt.max()) <= reuse.br.cstride < reuse.br.G // 4


def test_float16_candidate_reuse_replayed_from_a_hip_graph_and_in_chunks():
    """the plan is a device-side decision: a captured iteration replays it; set_crops() on a cached refiner starts from a full pass"""
    D, H, W, B = 40, 64, 64, 4
    K, p0, target, lidar = _problem(D, H, W, 2 * B)
    first = {k: v[:B] for k, v in p0.items()}
    second = {k: v[B:] for k, v in p0.items()}
    rows = []
    for reuse in (False, True):
        rf = sdflabel_amd.BatchRefiner(_dec16(reuse), D, K, (H, W), B, lidar_cap=4096, device=DEV)
        rf.set_crops(first, target[:B], [lidar] * B)
        rf.capture()
        out = []
        for params in (first, second):
            rf.set_crops(params, target[:B], [lidar] * B)
            rf.optimize(25)
            out.append(N(rf.results()[0]))
        rows.append(out)
    assert np.array_equal(rows[0][0], rows[1][0]) and np.array_equal(rows[0][1], rows[1][1])


def test_float16_candidate_reuse_latent_jump_forces_a_full_pass_for_that_crop_only():
    D, H, W, B = 40, 48, 48, 3
    K, p0, target, lidar = _problem(D, H, W, B)
    rf = sdflabel_amd.BatchRefiner(_dec16(True), D, K, (H, W), B, lidar_cap=4096, device=DEV)
    rf.set_crops(p0, target, [lidar] * B)
    rf.iteration(); rf.iteration()
    assert N(rf.br.reuse_flag).tolist() == [1, 1, 1]
    with torch.no_grad():
        rf.latent[1] += torch.tensor([0.5, -0.4, 0.3], device=DEV)
    rf.iteration()
    assert N(rf.br.reuse_flag).tolist() == [1, 0, 1]
    # the band of the moved crop is the full-grid band of its new latent
    chk = sdflabel_amd.BatchRenderer(_dec16(False), D, K, (W, H), B, device=DEV)
    # (the solver has stepped since: evaluate the parameters the renderer saw -- br.inputs holds the normalised latent rows of that step)
    chk.forward(rf.br.yaw, rf.br.trans, rf.br.latent)
    chk2 = sdflabel_amd.BatchRenderer(_dec16(True), D, K, (W, H), B, device=DEV)
    chk2.forward(rf.br.yaw, rf.br.trans, rf.br.latent)
    _ = chk.backward(g_color=torch.ones_like(chk.color)); _ = chk2.backward(g_color=torch.ones_like(chk2.color))
    _same_step(chk, chk2)


@pytest.mark.parametrize("precision", [torch.float16, torch.float32])
def test_candidate_reuse_audit_catches_a_band_row_outside_the_candidates(precision):
    """plant the failure the proof excludes: a true band row whose full-pass value is overwritten with 0.5 never becomes a candidate, so the
    candidate pass cannot bring it back -- the rotating audit (1/32 of the other rows per step, the mode's own kernel) must find it within one
    rotation and check_overflow() must refuse the result; without the audit it silently stays out of the band"""
    B, D, H, W = 2, 40, 32, 32
    a = [T(np.array([0.6, -0.4], np.float32)), T(np.array([[0.0, 0.0, 3.5], [0.1, -0.05, 3.2]], np.float32)),
         T(np.array([[0.3, -0.5, 0.8], [-0.2, 0.6, 0.4]], np.float32))]
    outcomes = {}
    for audit in (True, False):
        br = sdflabel_amd.BatchRenderer(_dec16(True, precision, candidate_audit=audit), D, K_for(H, W), (W, H), B, device=DEV)
        out = br.forward(*a)
        n1 = int(out["n"][1])
        g = int(br.idx[1, n1 // 2])
        br.fault = (torch.tensor([br.G + g], device=DEV), torch.tensor([0.5], device=DEV))
        br.invalidate_shape()                                          # the next step is a full pass: the fault enters the candidate selection
        caught = None
        for step in range(br.audit_stride if audit else 32):         # one whole rotation of the audit
            br.forward()
            assert int(br.cnt[1]) == n1 - 1
            if int(br.violations[1, 1]) > 0:
                caught = step
                break
        outcomes[audit] = caught
        assert int(br.violations[0].sum()) == 0
        if audit:
            assert caught is not None
            with pytest.raises(sdflabel_amd.SdfrError, match="candidate reuse"):
                br.check_overflow()
            br.fault = None
            br.forward(*a)
            for _ in range(br.audit_stride + 1):
                br.forward()
            assert br.prefilter_report()["hard_violations"] == 0 and int(br.cnt[1]) == n1
            br.check_overflow()
    assert outcomes[False] is None


@pytest.mark.parametrize("half", [True, False])
def test_ragged_forward_gives_each_row_the_bits_of_the_full_grid_launch(half):
    """C ABI: sdfr_mlp_forward_f16_ragged / sdfr_mlp_forward_ragged on gathered rows (two crops, counts that are no multiples of the tile, one
    empty crop) against sdfr_mlp_forward_f16 / sdfr_mlp_forward over all rows; rows beyond a crop's last 128-row block stay untouched"""
    L = _lib.lib()
    d = _dec16(False, torch.float16 if half else torch.float32)
    full_fn, ragged_fn = (L.sdfr_mlp_forward_f16, L.sdfr_mlp_forward_f16_ragged) if half else (L.sdfr_mlp_forward, L.sdfr_mlp_forward_ragged)
    h = d.handle(torch.device(DEV, 0))
    G, NI, B, stride = 5000, 6, 3, 1024
    gen = torch.Generator().manual_seed(3)
    x = (torch.rand(B * G, NI, generator=gen) * 2 - 1).to(DEV)
    full = torch.empty(B * G, device=DEV)
    _lib.check(full_fn(h.h, _lib.ptr(x), B * G, _lib.ptr(full), None, _lib.stream_ptr()), "fwd")
    cnt = torch.tensor([777, 0, 130], dtype=torch.int32, device=DEV)
    cidx = torch.zeros(B, stride, dtype=torch.int32, device=DEV)
    for b, n in enumerate(cnt.tolist()):
        cidx[b, :n] = torch.randperm(G, generator=gen)[:n].sort()[0].to(torch.int32).to(DEV)
    rows = torch.full((B * stride, NI), float("nan"), device=DEV)
    _lib.check(L.sdfr_candidate_rows(_lib.ptr(x), G, NI, B, _lib.ptr(cidx), stride, _lib.ptr(cnt), _lib.ptr(rows), _lib.stream_ptr()), "rows")
    out = torch.full((B * stride,), 7.0, device=DEV)
    masks = torch.zeros(int(L.sdfr_decoder_mask_words(h.h, B * stride)), dtype=torch.int32, device=DEV)
    _lib.check(ragged_fn(h.h, _lib.ptr(rows), B, stride, _lib.ptr(cnt), _lib.ptr(out), _lib.ptr(masks), 0, _lib.stream_ptr()), "ragged")
    if not half:                                                # half-size tiles are a float16 option
        assert ragged_fn(h.h, _lib.ptr(rows), B, stride, _lib.ptr(cnt), _lib.ptr(out), _lib.ptr(masks), 1, _lib.stream_ptr()) == -3
        out = out.view(B, stride)
        for b, n in enumerate(cnt.tolist()):
            assert torch.equal(out[b, :n], full[b * G + cidx[b, :n].long()])
        return
    # ... and on half-size tiles (the geometry of one or two crops per launch): the same bits again, other mask layout
    out_h = torch.full((B * stride,), 7.0, device=DEV)
    masks_h = torch.zeros_like(masks)
    _lib.check(ragged_fn(h.h, _lib.ptr(rows), B, stride, _lib.ptr(cnt), _lib.ptr(out_h), _lib.ptr(masks_h), 1, _lib.stream_ptr()), "ragged half tiles")
    # the mask-fed Jacobian from either mask set, told which layout it reads: identical rows
    n0 = int(cnt[0])
    pos = torch.arange(n0, dtype=torch.int32, device=DEV).view(1, -1).contiguous()
    c1 = torch.tensor([n0], dtype=torch.int32, device=DEV)
    J = [torch.zeros(n0, NI, device=DEV) for _ in range(2)]
    sel = [torch.zeros(n0, device=DEV) for _ in range(2)]
    for k, (o_, m_, flag) in enumerate(((out, masks, 0), (out_h, masks_h, 32))):
        _lib.check(L.sdfr_mlp_jacobian(h.h, _lib.ptr(rows), stride, 1, _lib.ptr(pos), n0, _lib.ptr(c1), _lib.ptr(J[k]), _lib.ptr(sel[k]), _lib.ptr(o_),
                                       _lib.ptr(m_), (2 if half else 0) | flag, _lib.stream_ptr()), "jac")
    assert torch.equal(J[0], J[1]) and torch.equal(sel[0], sel[1]) and float(J[0].abs().sum()) > 0
    out, out_h = out.view(B, stride), out_h.view(B, stride)
    for b, n in enumerate(cnt.tolist()):
        assert torch.equal(out[b, :n], full[b * G + cidx[b, :n].long()])
        assert torch.equal(out_h[b, :n], out[b, :n])
        pad = (n + 127) // 128 * 128                            # (sdfr_candidate_rows fills finite rows up to here; the f32 tile is 64 rows)
        assert bool(torch.isfinite(out[b, :((n + 63) // 64 * 64 if not half else pad)]).all()) and bool((out[b, pad:] == 7.0).all())
    assert ragged_fn(h.h, _lib.ptr(rows), B, 1000, _lib.ptr(cnt), _lib.ptr(out), None, 0, _lib.stream_ptr()) != 0     # not a multiple of the tile


def test_product_optimizer_uses_candidate_reuse_by_default_and_returns_the_same_bits():
    """pipelines.optimizer.Optimizer with the reference's shipped float16 setup (refine_css.py:144-153): candidate_reuse=True is the default;
    60 iterations give bitwise the parameters of candidate_reuse=False, and most of them ran on the candidates alone"""
    from sdflabel_amd.pipelines import optimizer as OP
    from tests._util import gold
    z = gold("g8_optimizer.npz")
    D, H, W = int(z["D"]), int(z["H"]), int(z["W"])
    init = z["init"]
    dsdf, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt")                                # float16, as the reference's default
    dsdf = dsdf.to(DEV)
    grid = sdflabel_amd.Grid3D(D, DEV, torch.float16)
    K = T(z["K"]).half()
    out = {}
    for reuse in (True, False):
        OP.clear_refiner_cache()
        params = {"yaw": init[0:1].copy(), "trans": init[1:4].copy(), "scale": init[4:5].copy(), "latent": init[5:8].copy()}
        opt = OP.Optimizer(params, DEV, {"2d": 0.3, "3d": 0.5}) if reuse else OP.Optimizer(params, DEV, {"2d": 0.3, "3d": 0.5}, candidate_reuse=False)
        opt.optimize(60, T(z["nocs_target"]).half(), z["lidar"], dsdf, grid, K, (H, W))
        assert opt._refiner.br.creuse == reuse
        if reuse:
            assert int(opt._refiner.br.n_full[0]) <= 6          # (warm-up / capture passes included) of 60 iterations
        out[reuse] = np.concatenate([N(params[k]).reshape(-1) for k in ("yaw", "trans", "scale", "latent")])
    OP.clear_refiner_cache()
    assert np.array_equal(out[True], out[False])


def test_exact_float32_reuse_through_the_product_optimizer_reproduces_golden_G8c():
    """the parity path with candidate reuse (the Optimizer's default) against the reference Optimizer's own 60-iteration run"""
    from sdflabel_amd.pipelines import optimizer as OP
    from tests._util import gold
    z = gold("g8c_optimizer_60it.npz")
    D, H, W = int(z["D"]), int(z["H"]), int(z["W"])
    init = z["init"]
    dsdf, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=torch.float32)
    dsdf = dsdf.to(DEV)
    grid = sdflabel_amd.Grid3D(D, DEV)
    out = {}
    for reuse in (True, False):
        OP.clear_refiner_cache()
        params = {"yaw": init[0:1].copy(), "trans": init[1:4].copy(), "scale": init[4:5].copy(), "latent": init[5:8].copy()}
        opt = OP.Optimizer(params, DEV, {"2d": 0.3, "3d": 0.5}, candidate_reuse=reuse)
        opt.optimize(60, T(z["nocs_target"]), z["lidar"], dsdf, grid, T(z["K"]), (H, W))
        assert opt._refiner.br.creuse == reuse
        out[reuse] = np.concatenate([N(params[k]).reshape(-1) for k in ("yaw", "trans", "scale", "latent")])
    OP.clear_refiner_cache()
    assert np.array_equal(out[True], out[False])
    assert np.abs(out[True] - z["traj"][-1]).max() < 1e-3


def test_kernel_errors_budgeted_by_the_reuse_proof_against_a_float64_evaluation():
    """E32 = 1e-6 is ASSUMED by BatchRenderer (the exact-f32 kernel against exact arithmetic) and the half kernel's deviation is calibrated
    against the exact-f32 kernel: check both against Decoder.forward_float64 on the whole grid for several latents"""
    d32, d16 = _dec16(False, torch.float32), _dec16(False, torch.float16)
    br = sdflabel_amd.BatchRenderer(_dec16(True, torch.float32), 40, K_for(32, 32), (32, 32), 1, device=DEV)
    L = _lib.lib()
    G = br.G
    gen = torch.Generator().manual_seed(5)
    worst32 = worst16 = 0.0
    for _ in range(3):
        lat = torch.nn.functional.normalize(torch.randn(3, generator=gen), dim=0).to(DEV)
        inp = torch.cat([lat.expand(G, -1), br.grid], 1).contiguous()
        ref = d32.forward_float64(inp).view(-1)
        s32, s16 = torch.empty(G, device=DEV), torch.empty(G, device=DEV)
        _lib.check(L.sdfr_mlp_forward(d32.handle(torch.device(DEV, 0)).h, _lib.ptr(inp), G, _lib.ptr(s32), None, _lib.stream_ptr()), "f32")
        _lib.check(L.sdfr_mlp_forward_f16(d16.handle(torch.device(DEV, 0)).h, _lib.ptr(inp), G, _lib.ptr(s16), None, _lib.stream_ptr()), "f16")
        worst32 = max(worst32, float((s32.double() - ref).abs().max()))
        worst16 = max(worst16, float((s16.double() - ref).abs().max()))
    assert worst32 < 5e-7 < 1e-6                                   # measured 1.6e-7
    assert worst16 <= 1.5 * br.select_error and br.margin >= 4 * br.select_error


# ---- r06: reuse off the shipped decoders ---------------------------------------------------------------------------------

def _amplify_a_dead_unit(d, want_bound=5e4):
    """The fixture decoder with a latent-Lipschitz BOUND >= want_bound (100x the shipped 532) and nearly the same function: hidden unit j of
    layer 1 loses its outgoing weights in layer 2 (one of 512 units) and has its own row scaled up, so ||W_1||_2 -- a factor of the product of
    spectral norms the proven bound is made of -- grows by two orders of magnitude while nothing downstream sees the unit."""
    j = 7
    with torch.no_grad():
        d.lin2.weight_v[:, j] = 0.0
        g = float(d.lin1.weight_g.view(-1)[j].abs())
        c = 1.25 * want_bound / d.latent_lipschitz_bound() * float(np.linalg.norm(d.effective_layers()[1][0], 2)) / g
        d.lin1.weight_g.view(-1)[j] *= c
        d.lin1.bias[j] *= c
    return d


@pytest.mark.parametrize("precision", [torch.float16, torch.float32])
def test_a_decoder_with_a_huge_lipschitz_bound_gets_a_full_pass_every_step_and_the_same_bits(precision):
    """the plan kernel's decision with a bound 100x the shipped decoder's: the latent moves ~1e-5 per iteration in normalised space, the candidate
    set is provably valid for |dz| <= margin / (4 lip) ~ 1e-8 only -> EVERY step runs the full grid; nothing raises and every bit equals the
    no-reuse evaluation (the fallback the design promises, never exercised by the shipped decoders)"""
    D, H, W, B, iters = 40, 48, 48, 3, 12
    K, p0, target, lidar = _problem(D, H, W, B)
    plain = sdflabel_amd.BatchRefiner(_amplify_a_dead_unit(_dec16(False, precision)), D, K, (H, W), B, lidar_cap=4096, device=DEV)
    reuse = sdflabel_amd.BatchRefiner(_amplify_a_dead_unit(_dec16(True, precision)), D, K, (H, W), B, lidar_cap=4096, device=DEV)
    assert reuse.br.creuse and reuse.br.lipschitz >= 5e4, reuse.br.lipschitz
    plain.set_crops(p0, target, [lidar] * B)
    reuse.set_crops(p0, target, [lidar] * B)
    br, reused = reuse.br, 0
    for it in range(iters):
        plain.iteration()
        reuse.iteration()
        _same_step(plain.br, br)
        # a crop may reuse its candidates only while the bound allows it: lip |z - z_ref| <= margin / 4, i.e. the (normalised) latent has moved
        # less than ~5e-8 since its last full pass -- a step the solver skipped for that crop (optimizer.py:127-129,149-151), nothing else
        flags = br.reuse_flag.bool()
        moved = (br.inputs.view(B, br.G, br.NI)[:, 0, :br.L] - br.lat_ref).norm(dim=1)
        assert bool((moved[flags] * br.lipschitz_plan <= br.margin / 4).all()) and bool((moved[flags] < 1e-7).all()), (it, moved, flags)
        reused += int(flags.sum())
    assert reused <= B * iters // 4, reused                               # (nearly) every step ran the whole grid
    assert int(plain.br.cnt.min()) > 200                                 # (the amplified decoder still has a shape)
    assert torch.equal(plain.results()[0], reuse.results()[0])          # results() -> check_overflow(): no exception
    rep = br.prefilter_report()
    assert rep["hard_violations"] == 0 and sum(rep["full_grid_passes_per_crop"]) == B * iters - reused and rep["lipschitz_bound"] >= 5e4


def test_a_kernel_error_beyond_the_margin_grows_the_margin_or_turns_reuse_off():
    """construction-time calibration: margin >= 4 x (safety x sampled half-kernel deviation + E32).  (a) a decoder.prefilter_margin below that is
    raised to it (reported); (b) when even decoder.candidate_max_margin is below it, reuse switches itself off -- the renderer evaluates the
    whole grid every step, says why in prefilter_report(), and gives the bits of a renderer that was never asked to reuse"""
    D, H, W, B = 40, 48, 48, 2
    K, p0, target, lidar = _problem(D, H, W, B)
    grown = sdflabel_amd.BatchRefiner(_dec16(True, prefilter_margin=1e-4), D, K, (H, W), B, lidar_cap=4096, device=DEV)
    br = grown.br
    assert br.creuse and br.margin_grown and br.margin == pytest.approx(4.0 * br.select_error) and br.margin > 1e-4
    assert br.f16_error == pytest.approx(2.0 * br.f16_deviation_sampled + br.e32)           # the safety factor on the sample maximum
    assert 1e-6 <= br.e32 < 2e-6 and 0 < sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=torch.float32)[0].to(DEV).kernel_error_f32(DEV) < 5e-7
    rep = br.prefilter_report()
    assert rep["candidate_reuse"] and rep["margin_grown_by_calibration"] and rep["margin"] == pytest.approx(br.margin)
    off = sdflabel_amd.BatchRefiner(_dec16(True, candidate_max_margin=1e-3), D, K, (H, W), B, lidar_cap=4096, device=DEV)
    plain = sdflabel_amd.BatchRefiner(_dec16(False), D, K, (H, W), B, lidar_cap=4096, device=DEV)
    assert not off.br.creuse and not off.br.guarded
    rep = off.br.prefilter_report()
    assert rep["candidate_reuse"] is False and "candidate_max_margin" in rep["reason"]
    assert plain.br.prefilter_report() is None
    rows = []
    for rf in (grown, off, plain):
        rf.set_crops(p0, target, [lidar] * B)
        rf.optimize(10)
        rows.append(N(rf.results()[0]))
    assert np.array_equal(rows[0], rows[2]) and np.array_equal(rows[1], rows[2])
