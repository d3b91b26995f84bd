"""GPU tests of the drop-in boundary's behaviour (round-2 review items): state hand-over through the autograd graph, empty bands, surfel
capacity overflow, the Optimizer mirror called exactly as the reference's pipeline calls it (float16 default precision), module names of
the reference's renderer package."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import sdflabel_amd
from sdflabel_amd import _lib
from tests._util import ASSET, K_for, gold
from tests.test_gpu_parity import N, T, build_pose

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def dec():
    d, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=torch.float32)
    return d.to(DEV)


def _inputs(grid, lat):
    lat_ = F.normalize(lat, p=2, dim=0)
    return torch.cat([lat_.expand(grid.points.size(0), -1), grid.points], 1)


@pytest.mark.parametrize("how", ["clone", "to_half_and_back", "view", "contiguous_float"])
def test_fused_band_path_survives_value_preserving_ops(dec, how):
    """`.clone()` / `.to()` / `.view()` between dsdf() and get_surface_points() must not fall to the generic path (which differentiates all
    G rows): the decoder state is found through the autograd graph, not through an attribute of the tensor object."""
    from sdflabel_amd.deepsdf.networks.deep_sdf_decoder_scale import sdf_state_of
    grid = sdflabel_amd.Grid3D(16, DEV)
    lat = torch.tensor([0.3, -0.5, 0.8], device=DEV, requires_grad=True)
    sdf, _ = dec(_inputs(grid, lat))
    ref_pts, _, ref_nrm = grid.get_surface_points(sdf)
    st = sdf_state_of(sdf)
    assert st is not None and st.J is not None
    lat2 = lat.detach().clone().requires_grad_(True)
    sdf2, _ = dec(_inputs(grid, lat2))
    moved = {"clone": lambda t: t.clone(), "to_half_and_back": lambda t: t.to(torch.float16).to(torch.float32),
             "view": lambda t: t.view(-1).view(-1, 1), "contiguous_float": lambda t: t.contiguous().float().clone()}[how](sdf2)
    assert not hasattr(moved, "_sdfr_state")
    st2 = sdf_state_of(moved)
    assert st2 is not None and st2.J is None
    pts, _, nrm = grid.get_surface_points(moved)
    assert st2.J is not None and st2.J.shape[0] == pts.shape[0]            # the band-only Jacobian ran (fused path)
    assert torch.equal(pts, ref_pts) and torch.equal(nrm, ref_nrm)
    pts.sum().backward()
    ref_pts.sum().backward()
    if how == "to_half_and_back":            # autograd rounds the gradient to half on its way back through the casts
        assert torch.allclose(lat.grad, lat2.grad, rtol=5e-3, atol=1e-4)
    else:
        assert torch.equal(lat.grad, lat2.grad)
    # a value-changing op is NOT followed: generic path, still correct
    lat3 = lat.detach().clone().requires_grad_(True)
    sdf3, _ = dec(_inputs(grid, lat3))
    assert sdf_state_of(sdf3 * 1.0) is None
    p3, _, _ = grid.get_surface_points(sdf3 * 1.0)
    assert np.abs(N(p3) - N(ref_pts)).max() < 1e-5


def test_second_consumer_of_the_decoder_output_keeps_its_gradient(dec):
    """the sync-free fast path of the decoder's backward trusts a gradient that says "band rows only".  With a second consumer
    of the decoder output (a regulariser on sdf) autograd adds the two gradients -- the sum must not pass as band-only (its out-of-band
    rows would be dropped), in either order of the two consumers; nor may a gradient built for an earlier band cache of the same state
    (get_surface_points called twice on one decoder output).  Checked against the gradients of the pieces taken separately."""
    grid = sdflabel_amd.Grid3D(16, DEV)

    def grad_of(build):
        lat = torch.tensor([0.3, -0.5, 0.8], device=DEV, requires_grad=True)
        grid.points.grad = None
        sdf, _ = dec(_inputs(grid, lat))
        build(sdf).backward()
        return lat.grad.clone(), grid.points.grad.clone()

    w = torch.linspace(-1.0, 2.0, grid.points.size(0), device=DEV).view(-1, 1)         # non-zero on every row, band or not
    reg = lambda sdf: (sdf * w).sum()
    surf = lambda sdf: (grid.get_surface_points(sdf)[0] * torch.tensor([1.0, -2.0, 0.5], device=DEV)).sum()
    wide = lambda sdf: grid.get_surface_points(sdf, threshold=0.06)[0].sum()
    g_reg, g_surf, g_wide = grad_of(reg), grad_of(surf), grad_of(wide)
    assert float(g_reg[0].abs().max()) > 1e-3 and float(g_surf[0].abs().max()) > 1e-3
    for combo, parts in ((lambda s: reg(s) + surf(s), (g_reg, g_surf)), (lambda s: surf(s) + reg(s), (g_reg, g_surf)),
                         (lambda s: surf(s) + wide(s), (g_surf, g_wide)), (lambda s: wide(s) + surf(s) + reg(s), (g_wide, g_surf, g_reg))):
        got = grad_of(combo)
        for k in range(2):
            want = sum(p_[k] for p_ in parts)
            assert torch.allclose(got[k], want, rtol=1e-4, atol=1e-5 * float(want.abs().max())), (k, (got[k] - want).abs().max())
    # the fast path itself still runs for the plain loop (one consumer): its tag vouches for the very tensor the decoder backward receives
    from sdflabel_amd.deepsdf.networks.deep_sdf_decoder_scale import BandTag
    seen = {}
    lat = torch.tensor([0.3, -0.5, 0.8], device=DEV, requires_grad=True)
    sdf, _ = dec(_inputs(grid, lat))
    sdf.register_hook(lambda g: seen.update(tag=getattr(g, "_sdfr_band_of", None)))          # (returns None: the gradient passes unchanged)
    surf(sdf).backward()
    assert isinstance(seen["tag"], BandTag)


def test_empty_band_gives_empty_tensors_and_zero_gradients(dec):
    """optimizer.py:127: the caller checks nelement() == 0; backward through an empty selection must give zeros, not raise"""
    grid = sdflabel_amd.Grid3D(8, DEV)
    lat = torch.tensor([0.3, -0.5, 0.8], device=DEV, requires_grad=True)
    yaw = torch.tensor([0.6], device=DEV, requires_grad=True)
    trans = torch.tensor([0.0, 0.0, 3.5], device=DEV, requires_grad=True)
    sdf, _ = dec(_inputs(grid, lat))
    pts, nocs, nrm = grid.get_surface_points(sdf, threshold=1e-12)
    assert pts.shape == (0, 3) and nocs.shape == (0, 3) and nrm.shape == (0, 3)
    r = sdflabel_amd.Rasterer(T(K_for(16, 16)), (16, 16)).to(DEV)
    rend, points = r(pts, nrm, nrm, build_pose(yaw, trans), rot="dcm", output_mask=True, output_nocs=True)
    assert points["xyzf"].nelement() == 0 and float(rend["color"].abs().max()) == 0.0
    (rend["color"].sum() + rend["mask"].sum() + points["xyzf"].sum() + pts.sum() + nocs.sum()).backward()
    assert lat.grad is not None and float(lat.grad.abs().max()) == 0.0
    assert float(yaw.grad.abs().max()) == 0.0 and float(trans.grad.abs().max()) == 0.0


def test_surfel_capacity_overflow_raises_in_the_refinement_loop(dec):
    z = gold("g8_optimizer.npz")
    D, H, W = int(z["D"]), int(z["H"]), int(z["W"])
    init = z["init"]
    rf = sdflabel_amd.BatchRefiner(dec, D, z["K"], (H, W), 1, lidar_cap=256, cap=64, device=DEV)
    rf.set_crops({"yaw": init[0:1], "trans": init[1:4][None], "scale": init[4:5], "latent": init[5:8][None]}, z["nocs_target"][None], [z["lidar"]])
    rf.optimize(2)
    with pytest.raises(_lib.SdfrError, match="cap"):
        rf.results()


def test_float16_refinement_against_the_references_own_float16_trajectory():
    """configs[4]'s precision through the whole refinement loop: the reference Optimizer run with its shipped float16 setup (golden G8h:
    half decoder, grid, K and target) beside its float32 run of the same problem (G8).  Tolerance stated up front, per parameter:
    tol = 2 * d_ref + 1e-4 with d_ref = the largest gap between the reference's own two precisions over the 10 iterations
    (2-3e-4 here) -- the float16 product path must stay within tol of BOTH reference trajectories at every iteration.  The Optimizer is
    called ten times for one iteration each, as the golden was recorded: the Adam state must carry over between the calls."""
    from sdflabel_amd.pipelines.optimizer import Optimizer
    z, zh = gold("g8_optimizer.npz"), gold("g8h_optimizer_fp16.npz")
    assert np.array_equal(z["nocs_target"], zh["nocs_target"]) and np.array_equal(z["init"], zh["init"])
    D, H, W = int(z["D"]), int(z["H"]), int(z["W"])
    init = z["init"]
    d_ref = np.abs(z["traj"] - zh["traj"]).max(axis=0)
    tol = 2 * d_ref + 1e-4
    dsdf, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt")                                # reference default: float16
    dsdf = dsdf.to(DEV)
    grid = sdflabel_amd.Grid3D(D, DEV, torch.float16)
    K = T(z["K"]).half()
    params = {"yaw": init[0:1].copy(), "trans": init[1:4].copy(), "scale": init[4:5].copy(), "latent": init[5:8].copy()}
    opt = Optimizer(params, DEV, {"2d": 0.3, "3d": 0.5})
    traj = []
    for _ in range(10):
        opt.optimize(1, T(z["nocs_target"]).half(), z["lidar"], dsdf, grid, K, (H, W))
        traj.append(np.concatenate([N(params[k]).reshape(-1) for k in ("yaw", "trans", "scale", "latent")]))
    traj = np.asarray(traj)
    e16, e32 = np.abs(traj - zh["traj"]).max(axis=0), np.abs(traj - z["traj"]).max(axis=0)
    assert (e16 <= tol).all(), (e16, tol)
    assert (e32 <= tol).all(), (e32, tol)


def test_optimizer_mirror_with_the_reference_pipelines_float16_setup():
    """refine_css.py:144-153 with the shipped config (precision = float16): setup_dsdf default precision, Grid3D(D, device, float16), half K.
    The mirror must accept the half grid and refine towards the float32 trajectory's end state."""
    from sdflabel_amd.pipelines.optimizer import Optimizer
    z = gold("g8_optimizer.npz")
    D, H, W = int(z["D"]), int(z["H"]), int(z["W"])
    init = z["init"]
    precision = torch.float16
    dsdf, latent_size = sdflabel_amd.setup_dsdf(ASSET + ".pt")                      # reference default: float16
    assert dsdf.mlp_precision == torch.float16
    dsdf = dsdf.to(DEV)
    grid = sdflabel_amd.Grid3D(D, DEV, precision)
    K = T(z["K"]).to(precision)
    params = {"yaw": init[0:1].copy(), "trans": init[1:4].copy(), "scale": init[4:5].copy(), "latent": init[5:8].copy()}
    opt = Optimizer(params, DEV, {"2d": 0.3, "3d": 0.5})
    opt.optimize(10, T(z["nocs_target"]).to(precision), z["lidar"], dsdf, grid, K, (H, W))
    got = np.concatenate([N(params[k]).reshape(-1) for k in ("yaw", "trans", "scale", "latent")])
    assert np.isfinite(got).all()
    assert np.abs(got - z["traj"][-1]).max() < 2e-2, np.abs(got - z["traj"][-1])       # float16 decoder: close to the float32 trajectory
    assert abs(got[0] - init[0]) > 0.05
    # a second Optimizer object (the reference constructs one per crop, refine_css.py:203) reuses the refiner and its captured graph
    opt2 = Optimizer({k: np.asarray(v.detach().cpu()) for k, v in params.items()}, DEV, {"2d": 0.3, "3d": 0.5})
    opt2.optimize(5, T(z["nocs_target"]).to(precision), z["lidar"], dsdf, grid, K, (H, W))
    assert opt2._refiner is opt._refiner and opt._refiner._replay is not None
    # a grid that is not the D^3 point set is refused
    bad = sdflabel_amd.Grid3D(D, DEV, precision)
    with torch.no_grad():
        bad.points[5, 0] += 0.25
    with pytest.raises(_lib.SdfrError):
        Optimizer({k: np.asarray(v.detach().cpu()) for k, v in params.items()}, DEV, {"2d": 0.3, "3d": 0.5}).optimize(
            2, T(z["nocs_target"]), z["lidar"], dsdf, bad, K, (H, W))


def test_rasterer_state_dict_keys_match_the_reference():
    r = sdflabel_amd.Rasterer(T(K_for(32, 24)), (32, 24))
    assert list(r.state_dict().keys()) == ["grid", "grid_prim", "K"]              # rasterer.py:27,32,44
    assert r.grid_prim.shape == (1, 225, 2) and r.grid.shape == (1, 32 * 24, 2)


# ---- the reference's renderer.primitives / renderer.projection module functions (golden G13, captured from the reference) ------------

def _call_primitive(name, bg, K, r, uv, p, n):
    from sdflabel_amd.renderer import primitives as prim
    if name == "disc":
        return prim.inside_surfel(K, r.grid, uv, p, n, diam=0.04, softclamp=False, add_bg=bg)
    if name == "circle":
        return prim.inside_circle(K, r.grid, uv, p, n, diam=0.02, add_bg=bg)
    return prim.inside_circle_opt(K, r.grid_prim, uv, p, n, diam=0.025, add_bg=bg)


@pytest.mark.parametrize("name", ["disc", "circle", "circle_opt"])
@pytest.mark.parametrize("bg", [False, True])
def test_standalone_primitives_dense_weights_and_gradients_golden(name, bg):
    z = gold("g13_primitives.npz")
    W, H = [int(v) for v in z["res"]]
    K = T(z["K"])
    r = sdflabel_amd.Rasterer(K, (W, H)).to(DEV)
    p = T(z["points"]).requires_grad_(True)
    n = T(z["normals"]).requires_grad_(True)
    t = "%s_bg%d_" % (name, int(bg))
    w = _call_primitive(name, bg, K, r, T(z["uv"]), p, n)
    ref = z[t + "w"]
    assert w.shape == (ref.shape[0], 3, W * H)
    assert torch.equal(w[:, 0], w[:, 1]) and torch.equal(w[:, 0], w[:, 2])
    tol = 1e-3 if name == "circle_opt" else 1e-5          # circle_opt: logits scaled by 10000, one float32 ulp moves a weight by ~2e-4
    assert np.abs(N(w[:, 0]) - ref).max() < tol
    assert ((N(w[:, 0]) > 0) == (ref > 0)).all()
    (w[:, 0, :] * T(z[t + "R"])).sum().backward()
    gtol = 2e-2 if name == "circle_opt" else 2e-3
    for got, key in ((p.grad, "g_points"), (n.grad, "g_normals")):
        g = z[t + key]
        got = torch.zeros_like(p) if got is None else got
        assert np.abs(N(got) - g).max() < gtol * max(1.0, np.abs(g).max()), (key, np.abs(N(got) - g).max(), np.abs(g).max())


def test_standalone_primitives_refuse_only_what_is_no_primitive():
    from sdflabel_amd.renderer import primitives as prim
    K = T(K_for(16, 16))
    r = sdflabel_amd.Rasterer(K, (16, 16)).to(DEV)
    p = torch.rand(4, 3, device=DEV) + torch.tensor([0, 0, 1.0], device=DEV)
    w = prim.inside_surfel(K, r.grid, None, p, p)                                  # the function's own defaults (softclamp=True, add_bg=True): built since r05
    assert w.shape == (5, 3, 256) and bool(torch.isfinite(w).all())
    with pytest.raises(NotImplementedError):
        prim.inside_surfel(K, r.grid[:, ::2], None, p, p, softclamp=False)         # not the full pixel grid
    with pytest.raises(NotImplementedError):
        prim.inside_circle(K, r.grid, p[:, :2], p, p, softclamp_constant=-1.0)     # a sigmoid that covers what is FAR from the vertex


G13S = {   # case -> (function, keyword arguments): the reference calls behind golden G13s (tools/make_golden.py g13s)
    "disc_default_bg1": ("inside_surfel", dict()),
    "disc_soft_bg0": ("inside_surfel", dict(diam=0.04, softclamp=True, add_bg=False)),
    "disc_soft_c40_bg1": ("inside_surfel", dict(diam=0.04, softclamp=True, softclamp_constant=40, add_bg=True)),
    "circle_hard_bg0": ("inside_circle", dict(diam=0.02, softclamp=False, add_bg=False)),
    "circle_hard_bg1": ("inside_circle", dict(diam=0.02, softclamp=False, add_bg=True)),
    "circle_c30_default_diam_bg0": ("inside_circle", dict(softclamp_constant=30)),
    "circle_opt_hard_bg0": ("inside_circle_opt", dict(diam=0.025, softclamp=False, add_bg=False)),
    "circle_opt_hard_bg1": ("inside_circle_opt", dict(diam=0.025, softclamp=False, add_bg=True)),
}


@pytest.mark.parametrize("case", sorted(G13S))
def test_standalone_primitives_other_clamp_configurations_golden(case):
    """VERDICT r04 missing 4: the clamp configurations Rasterer.forward never passes -- inside_surfel's own defaults (softclamp=True, add_bg=True:
    a dense problem, the sigmoid mask holds until exp overflows), the hard-edged circles, another softclamp_constant -- against the
    reference's weights and autograd gradients (golden G13s)"""
    from sdflabel_amd.renderer import primitives as prim
    z, zs = gold("g13_primitives.npz"), gold("g13s_primitive_clamps.npz")
    W, H = [int(v) for v in z["res"]]
    K = T(z["K"])
    r = sdflabel_amd.Rasterer(K, (W, H)).to(DEV)
    p = T(z["points"]).requires_grad_(True)
    n = T(z["normals"]).requires_grad_(True)
    fn, kw = G13S[case]
    grid = r.grid_prim if fn == "inside_circle_opt" else r.grid
    w = getattr(prim, fn)(K, grid, T(z["uv"]), p, n, **kw)
    ref = zs[case + "_w"]
    assert w.shape == (ref.shape[0], 3, W * H)
    assert torch.equal(w[:, 0], w[:, 1]) and torch.equal(w[:, 0], w[:, 2])
    tol = 1e-3 if fn == "inside_circle_opt" else 1e-5
    assert np.abs(N(w[:, 0]) - ref).max() < tol
    # the same (surfel, pixel) pairs carry weight -- but for weights in the denormal range, which exp() flushes to 0 on one side only (the soft
    # disc's softmax runs over all N surfels of a pixel: logits 150 apart give e^-100)
    got = N(w[:, 0])
    assert (((got > 0) == (ref > 0)) | (np.maximum(got, ref) < 1e-30)).all()
    (w[:, 0, :] * T(zs[case + "_R"])).sum().backward()
    gtol = 2e-2 if fn == "inside_circle_opt" else 2e-3
    for got, key in ((p.grad, "_g_points"), (n.grad, "_g_normals")):
        g = zs[case + key]
        got = torch.zeros_like(p) if got is None else got
        assert np.abs(N(got) - g).max() < gtol * max(1.0, np.abs(g).max()), (key, np.abs(N(got) - g).max(), np.abs(g).max())


@pytest.mark.parametrize("tag", ["dcm", "quat"])
def test_standalone_projection_outputs_and_gradients_golden(tag):
    from sdflabel_amd.renderer import projection as proj
    z = gold("g13_primitives.npz")
    K = T(z["proj_K"])
    p = T(z["proj_points"]).requires_grad_(True)
    n = T(z["proj_normals"]).requires_grad_(True)
    cam = T(z["proj_%s_cam" % tag]).requires_grad_(True)
    fn = proj.project_in_2D if tag == "dcm" else proj.project_in_2D_quat
    o = fn(K, cam, p, n, n, (32, 32), output_nocs=True)
    keys = sorted(k[len("proj_%s_out_" % tag):] for k in z.files if k.startswith("proj_%s_out_" % tag))
    assert sorted(o) == keys
    loss = 0
    for k in keys:
        ref = z["proj_%s_out_%s" % (tag, k)]
        assert o[k].shape == ref.shape, k
        assert np.abs(N(o[k]) - ref).max() < 2e-5 * max(1.0, np.abs(ref).max()), k
        loss = loss + (o[k] * T(z["proj_%s_W_%s" % (tag, k)])).sum()
    loss.backward()
    for got, key in ((cam.grad, "g_cam"), (p.grad, "g_points"), (n.grad, "g_normals")):
        ref = z["proj_%s_%s" % (tag, key)]
        if tag == "dcm" and key == "g_cam":
            got, ref = got[:3], ref[:3]                  # the homogeneous row is dropped (projection.py:34)
        assert np.abs(N(got) - ref).max() < 1e-3 * max(1.0, np.abs(ref).max()), (key, np.abs(N(got) - ref).max())
    with pytest.raises(NotImplementedError):
        fn(K, cam, p, n, n, (32, 32), filter_hpr=True)


def test_reference_module_names_resolve_through_the_compat_path():
    import importlib
    import os
    import sys
    compat = os.path.join(os.path.dirname(sdflabel_amd.__file__), "compat")
    sys.path.insert(0, compat)
    try:
        for mod, names in (("renderer.rasterer", ["Rasterer"]), ("renderer.primitives", ["inside_circle", "inside_circle_opt", "inside_surfel"]),
                           ("renderer.projection", ["project_in_2D", "project_in_2D_quat"]), ("grid", ["Grid3D"]),
                           ("sdfrenderer.grid", ["Grid3D"]), ("deepsdf.workspace", ["setup_dsdf"]),
                           ("sdfrenderer.deepsdf.workspace", ["setup_dsdf"]), ("sdfrenderer.renderer.primitives", ["inside_surfel"])):
            m = importlib.import_module(mod)
            assert all(hasattr(m, n) for n in names), mod
    finally:
        sys.path.remove(compat)


@pytest.mark.parametrize("kw", [dict(latent_in=[2, 4], xyz_in_all=False), dict(latent_in=[3], xyz_in_all=True), dict(latent_in=(), xyz_in_all=True)])
def test_half_forward_k_side_injection_for_other_injection_patterns(kw):
    """the half forward keeps re-injected input columns behind the feature slots of its operand tile (K-side injection): check it on 512-wide
    decoders with several latent_in layers and with xyz_in_all (injection offset = latent size, injection in every layer) against the
    exact-f32 kernels of the same weights"""
    torch.manual_seed(5)
    d = sdflabel_amd.Decoder(3, dims=[300, 512, 512, 400, 512, 512], norm_layers=(), weight_norm=False, **kw)
    with torch.no_grad():
        for p in d.parameters():
            p.mul_(1.5)
    d = d.to(DEV).eval()
    x = (torch.randn(700, 6, device=DEV) * 0.6).contiguous()
    d.mlp_precision = torch.float32
    s32, _ = d(x)
    d.mlp_precision = torch.float16
    s16, _ = d(x)
    err = float((s32 - s16).abs().max())
    assert 0 < err < 1e-2, err
    # and the mask-fed half Jacobian on top of it stays close to the exact one
    xr = x.clone().requires_grad_(True)
    d.mlp_precision = torch.float32
    d(xr)[0].sum().backward()
    g32 = xr.grad.clone()
    xr.grad = None
    d.mlp_precision = torch.float16
    d(xr)[0].sum().backward()
    assert float((xr.grad - g32).abs().max()) < 5e-2 * max(1.0, float(g32.abs().max()))


def test_standalone_functions_with_no_surfels():
    """empty selections give empty / background-only results, as the reference's ops on (0,3) tensors do"""
    from sdflabel_amd.renderer import primitives as prim, projection as proj
    K = T(K_for(16, 16))
    r = sdflabel_amd.Rasterer(K, (16, 16)).to(DEV)
    e3, e2 = torch.zeros(0, 3, device=DEV), torch.zeros(0, 2, device=DEV)
    w = prim.inside_surfel(K, r.grid, e2, e3, e3, diam=0.04, softclamp=False, add_bg=False)
    assert w.shape == (0, 3, 256)
    wb = prim.inside_surfel(K, r.grid, e2, e3, e3, diam=0.04, softclamp=False, add_bg=True)
    assert wb.shape == (1, 3, 256) and float(wb.min()) == 1.0
    o = proj.project_in_2D(K, torch.eye(4, device=DEV), e3, e3, e3, (16, 16))
    assert o["points_3d"].shape == (0, 3) and o["points_3d_filt"].shape == (0, 3) and o["points_2d"].shape == (0, 2)


def test_rasterer_fused_fast_path_equals_the_general_path(dec=None):
    """the optimizer's configuration (rot='dcm', 'disc', bg=None, output_nocs=True) runs through _RasterDiscFn (fused projection, xyzf and rgbf
    rows from the kernel, one projection backward): same kernels, so the same bits as the general autograd.Function -- images, the four point
    outputs and every gradient, including those arriving through xyz / rgb / rgbf"""
    rng = np.random.default_rng(3)
    H, W, n = 40, 56, 700
    p = np.stack([rng.uniform(-0.5, 0.5, n), rng.uniform(-0.4, 0.4, n), rng.uniform(-0.9, 0.9, n)], 1).astype(np.float32)
    nr = rng.standard_normal((n, 3)).astype(np.float32)
    nr /= np.linalg.norm(nr, axis=1, keepdims=True)
    K = K_for(H, W)
    K[0, 2] -= 11.0
    res = []
    for fast in (True, False):
        r = sdflabel_amd.Rasterer(T(K), (W, H)).to(DEV)
        r.fast_path = fast
        pts, nrm = T(p).requires_grad_(True), T(nr).requires_grad_(True)
        yaw, trans = T(np.array([0.7], np.float32)).requires_grad_(True), T(np.array([0.1, -0.05, 3.2], np.float32)).requires_grad_(True)
        from tests.test_gpu_parity import build_pose
        rend, points = r(pts, nrm, nrm, build_pose(yaw, trans), primitives='disc', rot='dcm', bg=None, output_depth=True, output_normals=True,
                         output_nocs=True, output_points=True, output_mask=True)
        gen = torch.Generator().manual_seed(1)
        loss = sum((rend[k] * torch.randn(rend[k].shape, generator=gen).to(DEV)).sum() for k in ("color", "mask", "depth", "normals"))
        loss = loss + sum((points[k] * torch.randn(points[k].shape, generator=gen).to(DEV)).sum() for k in ("xyz", "rgb", "xyzf", "rgbf"))
        loss.backward()
        res.append(([rend[k].detach() for k in ("color", "mask", "depth", "normals")] + [points[k].detach() for k in ("xyz", "rgb", "xyzf", "rgbf")],
                    [pts.grad, nrm.grad, yaw.grad, trans.grad]))
    assert res[0][0][6].shape[0] > 50
    for a, b in zip(res[0][0], res[1][0]):
        assert torch.equal(a, b)
    for a, b in zip(res[0][1], res[1][1]):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6), float((a - b).abs().max())
    # empty input and an unused-output backward
    r = sdflabel_amd.Rasterer(T(K), (W, H)).to(DEV)
    e = torch.zeros((0, 3), device=DEV, requires_grad=True)
    rend, points = r(e, e, e, torch.eye(4, device=DEV), rot='dcm', output_nocs=True, output_mask=True)
    assert points['xyzf'].shape == (0, 3) and float(rend['color'].abs().sum()) == 0.0
    rend['color'].sum().backward()
    pts = T(p).requires_grad_(True)
    rend, points = r(pts, T(nr), T(nr), torch.eye(4, device=DEV) + torch.tensor([[0, 0, 0, 0], [0, 0, 0, 0], [0, 0, 0, 3.0], [0, 0, 0, 0]], device=DEV),
                     rot='dcm', output_nocs=True, output_mask=True)
    points['xyzf'].sum().backward()                            # only the xyzf gradient arrives: no image gradient at all
    assert bool(torch.isfinite(pts.grad).all()) and float(pts.grad.abs().sum()) > 0


def test_decoder_scale_head_fused_kernel_equals_the_torch_modules():
    """Decoder.forward returns (sdf, scale) as the reference does (deep_sdf_decoder_scale.py:110-114); the scale head runs as one launch
    (sdfr_scale_net) and must equal scale_net evaluated with torch ops, value and gradient w.r.t. the latent"""
    d, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=torch.float32)
    d = d.to(DEV)
    grid = sdflabel_amd.Grid3D(8, DEV)
    lat = torch.tensor([0.3, -0.5, 0.8], device=DEV, requires_grad=True)
    inp = torch.cat([lat.expand(grid.points.size(0), -1), grid.points], 1)
    sdf, scale = d(inp)
    ref = d.scale_net(lat)
    assert scale.shape == ref.shape and float((scale - ref).abs().max()) < 1e-6
    (g,) = torch.autograd.grad(scale.sum(), lat, retain_graph=True)
    (gr,) = torch.autograd.grad(ref.sum(), lat)
    assert float((g - gr).abs().max()) < 1e-6
    # the head's own parameters get their gradients too (fine-tuning the scale head through Decoder.forward), accumulated as
    # autograd accumulates into leaves
    params = list(d.scale_net.parameters())
    assert all(p_.requires_grad for p_ in params)
    for p_ in params:
        p_.grad = None
    d.scale_net(lat.detach()).sum().backward()
    want = [p_.grad.clone() for p_ in params]
    for p_ in params:
        p_.grad = None
    _, scale2 = d(inp.detach())
    scale2.sum().backward()
    scale3 = d(inp.detach())[1]
    scale3.sum().backward()                                     # a second backward accumulates
    for p_, w_ in zip(params, want):
        assert p_.grad is not None and float((p_.grad - 2 * w_).abs().max()) < 1e-6 * max(1.0, float(w_.abs().max()))
        p_.grad = None


def test_drop_in_iterations_do_not_accumulate_device_memory():
    """the reference's loop calls dsdf / get_surface_points / Rasterer / backward 60 times per crop: every buffer of an iteration must be
    released with its autograd graph.  (r03: the decoder state used to hold a VIEW of the decoder's output while the output held the state --
    a reference cycle through C++ that Python's collector cannot see: 36 MB leaked per iteration at D = 40, the decoder kernel 1.7x slower
    on never-touched pages.)"""
    import gc
    D, H, W = 40, 128, 128
    d, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=torch.float32)
    d = d.to("cuda")
    grid = sdflabel_amd.Grid3D(D, "cuda")
    renderer = sdflabel_amd.Rasterer(torch.from_numpy(K_for(H, W)), (W, H)).to("cuda")
    yaw = torch.tensor([0.6], device="cuda", requires_grad=True)
    trans = torch.tensor([0.05, -0.03, 3.5], device="cuda", requires_grad=True)
    lat = torch.tensor([0.3, -0.5, 0.8], device="cuda", requires_grad=True)

    def iteration(half=False):
        for p in (yaw, trans, lat):
            p.grad = None
        inputs = torch.cat([F.normalize(lat, p=2, dim=0).expand(grid.points.size(0), -1), grid.points], 1)
        sdf, _ = d(inputs.half() if half else inputs)
        pcd, nocs, normals = grid.get_surface_points(sdf.float())
        rendering, points = renderer(pcd, normals, normals, build_pose(yaw, trans), primitives='disc', rot='dcm', bg=None, output_depth=False,
                                     output_normals=True, output_nocs=True, output_points=True, output_mask=True)
        (rendering['color'].sum() + rendering['mask'].sum() + points['xyzf'].sum()).backward()

    gc.disable()                                  # nothing here may depend on the cycle collector
    try:
        for half in (False, True):                # (half inputs: the output is a converted copy carrying the same state)
            for _ in range(3):
                iteration(half)
            torch.cuda.synchronize()
            base = torch.cuda.memory_allocated()
            for _ in range(20):
                iteration(half)
            torch.cuda.synchronize()
            grown = torch.cuda.memory_allocated() - base
            assert grown < 4 << 20, "device memory grew by %.1f MB over 20 iterations (half inputs: %s)" % (grown / 1e6, half)
    finally:
        gc.enable()


def test_decoder_handle_is_bound_to_its_device_and_errors_are_reported_not_faults():
    """r06 (first-run hardening for the multi-GPU path): a decoder's weight images live on ONE device.  Creating a handle for a
    device the process cannot see returns a HIP error through sdfr_last_error (no abort), the caller's current device is left as it was, and every
    decoder launch checks that the current device is the handle's (on a box with >= 2 GPUs: a launch with the other device current is refused)."""
    d, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=torch.float32)
    d = d.to(DEV)
    n_dev = torch.cuda.device_count()
    before = torch.cuda.current_device()
    with pytest.raises(_lib.SdfrError, match="(?i)hip|device"):
        d.handle(torch.device("cuda", n_dev))                     # one past the last device
    assert torch.cuda.current_device() == before
    x = torch.zeros(64, 6, device=DEV)
    out = torch.empty(64, device=DEV)
    h = d.handle(torch.device("cuda", before))
    _lib.check(_lib.lib().sdfr_mlp_forward(h.h, _lib.ptr(x), 64, _lib.ptr(out), None, _lib.stream_ptr()), "sdfr_mlp_forward")
    if n_dev >= 2:
        other = (before + 1) % n_dev
        with torch.cuda.device(other):
            rc = _lib.lib().sdfr_mlp_forward(h.h, _lib.ptr(x), 64, _lib.ptr(out), None, _lib.stream_ptr())
        assert rc != 0 and "current device" in _lib.lib().sdfr_last_error().decode()
