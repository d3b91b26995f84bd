"""GPU parity tests: the HIP path (through the C ABI / the drop-in Python boundary) against the numpy oracle and the
golden vectors captured from the reference.  Tolerances: 1e-4 on rendered images (BASELINE.json north_star), tighter on
per-kernel quantities.  Discontinuous selections (band, front-face, disc) are exact on the fixtures, whose minimum margins to
the thresholds are recorded in the goldens; where a flip of a (pixel, surfel) pair within float rounding of the disc edge is
possible, the affected pixels must be attributable to a margin < 1e-5 and are bounded to 0.1 % of the image."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import sdflabel_amd
from sdflabel_amd import _lib
from oracle import sdf_oracle as O
from tests._util import ASSET, K_for, fitted_state, gold, state_from_npz

pytestmark = pytest.mark.gpu
DEV = "cuda"


def T(a, **kw):
    return torch.as_tensor(np.ascontiguousarray(a), device=DEV, **kw)


def N(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope="module")
def dec():
    d, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=torch.float32)
    return d.to(DEV)


@pytest.fixture(scope="module")
def oracle_layers():
    st, spec = fitted_state()
    return O.decoder_layers_from_state(st, spec), spec


def rot_from_yaw(yaw):
    c, s = torch.cos(yaw), torch.sin(yaw)
    z, o = yaw.new_zeros(1), yaw.new_ones(1)
    return torch.stack((c, z, s, z, o, z, -s, z, c)).view(3, 3)


def build_pose(yaw, trans):
    """the pose construction of pipelines/optimizer.py:86-90 (a9 harness)"""
    pose = torch.eye(4, device=yaw.device)
    pose[:3, :3] = rot_from_yaw(yaw)
    pose[1] *= -1
    pose[:3, 3] = trans
    return pose


def images_close(got, ref, aux=None, atol=1e-4, frac=1e-3):
    got, ref = np.asarray(got), np.asarray(ref)
    assert got.shape == ref.shape
    bad = np.abs(got - ref) > atol
    if not bad.any():
        return
    assert aux is not None, "max diff %g" % np.abs(got - ref).max()
    badpix = bad.reshape(bad.shape[0], -1).any(axis=0)
    near = (aux["margin_disc"] < 1e-5) | (aux["margin_b"] < 1e-5)
    assert not (badpix & ~near).any(), "pixels differ beyond tolerance away from any selection threshold"
    assert badpix.mean() <= frac


# ---- decoder ------------------------------------------------------------------------------------------------------

def test_mlp_forward_golden_fitted(dec):
    z = gold("g2_decoder.npz")
    inp = T(z["fit_inputs"])
    sdf, scale = dec(inp)
    assert sdf.shape == (inp.shape[0], 1)
    assert np.abs(N(sdf) - z["fit_sdf"]).max() < 5e-6
    assert np.allclose(N(scale), z["fit_scale"], atol=1e-6)


def test_mlp_backward_golden_fitted(dec):
    z = gold("g2_decoder.npz")
    inp = T(z["fit_inputs"]).requires_grad_(True)
    sdf, _ = dec(inp)
    sdf.sum().backward()          # generic backward: no band cache, every row carries gradient
    ref = z["fit_grad_inputs"]
    assert np.abs(N(inp.grad) - ref).max() < 2e-5 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("tag,kw", [
    ("ln", dict(latent_size=3, dims=[64] * 8, dropout=list(range(8)), dropout_prob=0.2, norm_layers=list(range(8)), latent_in=[4],
                weight_norm=False)),
    ("wn", dict(latent_size=3, dims=[64] * 8, dropout=list(range(8)), dropout_prob=0.2, norm_layers=list(range(8)), latent_in=[4],
                weight_norm=True)),
    ("x", dict(latent_size=5, dims=[48] * 5, dropout=None, norm_layers=(), latent_in=[2, 4], weight_norm=False, xyz_in_all=True,
               use_tanh=True)),
])
def test_mlp_small_specs_golden(tag, kw):
    z = gold("g2_decoder.npz")
    d = sdflabel_amd.Decoder(**kw)
    d.load_state_dict({k: torch.from_numpy(v) for k, v in state_from_npz(z, tag + "_state_").items()})
    d = d.to(DEV).eval()
    inp = T(z[tag + "_inputs"]).requires_grad_(True)
    sdf, scale = d(inp)
    assert np.abs(N(sdf) - z[tag + "_sdf"]).max() < 5e-6
    g_out = T(z[tag + "_gout"]) if (tag + "_gout") in z.files else torch.ones_like(sdf)
    (sdf * g_out).sum().backward()
    ref = z[tag + "_grad_inputs"]
    assert np.abs(N(inp.grad) - ref).max() < 2e-5 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("width", [200, 512])
def test_layernorm_decoder_wide_vs_oracle(width):
    """the LayerNorm variant (weight_norm=False, deep_sdf_decoder_scale.py:56-57,99-101) at the other padded widths: forward and
    input Jacobian against the oracle, then through Grid3D.get_surface_points (recomputing Jacobian with the x_hat scratch)."""
    torch.manual_seed(width)
    d = sdflabel_amd.Decoder(3, dims=[width] * 4, norm_layers=[0, 1, 2, 3], latent_in=[2], weight_norm=False)
    with torch.no_grad():
        for p in d.parameters():
            p.add_(0.05 * torch.randn_like(p))
    d = d.to(DEV).eval()
    st = {k: v.detach().cpu().numpy() for k, v in d.state_dict().items()}
    spec = dict(dims=[width] * 4, latent_in=[2])
    layers = O.decoder_layers_from_state(st, spec)
    rng = np.random.default_rng(width)
    inp = (rng.standard_normal((333, 6)) * 0.7).astype(np.float32)
    x = T(inp).requires_grad_(True)
    sdf, _ = d(x)
    ref, cache = O.decoder_forward(layers, spec, inp, want_cache=True)
    assert np.abs(N(sdf) - ref).max() < 1e-5
    g_out = rng.standard_normal(ref.shape).astype(np.float32)
    (sdf * T(g_out)).sum().backward()
    gref = O.decoder_backward_inputs(layers, spec, inp, cache, g_out)
    assert np.abs(N(x.grad) - gref).max() < 5e-5 * max(1.0, np.abs(gref).max())
    grid = sdflabel_amd.Grid3D(8, DEV)
    lat = torch.tensor([0.2, -0.1, 0.4], device=DEV)
    inputs = torch.cat([lat.expand(512, -1), grid.points], 1)
    s2, _ = d(inputs)
    pts, _, nrm = grid.get_surface_points(s2, threshold=10.0)
    r2, c2 = O.decoder_forward(layers, spec, N(inputs), want_cache=True)
    J = O.decoder_backward_inputs(layers, spec, N(inputs), c2, np.ones_like(r2))
    pm, _, nm, _, _ = O.get_surface_points(N(grid.points), r2, J[:, 3:], 10.0)
    assert pts.shape[0] == 512 and np.abs(N(pts) - pm).max() < 1e-4 and np.abs(N(nrm) - nm).max() < 1e-3


@pytest.mark.parametrize("n", [1, 63, 64, 65, 1000])
def test_mlp_forward_ragged_sizes_vs_oracle(dec, oracle_layers, n):
    layers, spec = oracle_layers
    rng = np.random.default_rng(n)
    inp = (rng.standard_normal((n, 6)) * 0.6).astype(np.float32)
    sdf, _ = dec(T(inp))
    ref = O.decoder_forward(layers, spec, inp)
    assert np.abs(N(sdf) - ref).max() < 5e-6


def test_mlp_jacobian_selected_rows_vs_oracle(dec, oracle_layers):
    layers, spec = oracle_layers
    rng = np.random.default_rng(5)
    inp = (rng.standard_normal((500, 6)) * 0.6).astype(np.float32)
    rows = np.sort(rng.choice(500, 77, replace=False)).astype(np.int32)
    from sdflabel_amd.deepsdf.networks.deep_sdf_decoder_scale import SdfState, mlp_jacobian
    st = SdfState(dec.handle(torch.device(DEV, 0)), T(inp))
    J, sel = mlp_jacobian(st, T(rows), len(rows))
    sdf, cache = O.decoder_forward(layers, spec, inp, want_cache=True)
    Jref = O.decoder_backward_inputs(layers, spec, inp, cache, np.ones_like(sdf))[rows]
    assert np.abs(N(sel) - sdf[rows, 0]).max() < 5e-6
    assert np.abs(N(J) - Jref).max() < 2e-5 * max(1.0, np.abs(Jref).max())


def test_mlp_jacobian_from_saved_masks_equals_recompute(dec):
    """The mask-fed (backward-only) Jacobian equals the recomputing one up to the summation order of the last 512->1 dot
    (8 vs 16 partial sums), i.e. to float rounding; the selected sdf values are exactly the forward launch's."""
    from sdflabel_amd.deepsdf.networks.deep_sdf_decoder_scale import mlp_jacobian
    rng = np.random.default_rng(11)
    inp = T((rng.standard_normal((3000, 6)) * 0.6).astype(np.float32))
    sdf, _ = dec(inp)
    st = sdf._sdfr_state
    rows = T(np.sort(rng.choice(3000, 517, replace=False)).astype(np.int32))
    J1, s1 = mlp_jacobian(st, rows, 517, use_masks=True)
    J2, s2 = mlp_jacobian(st, rows, 517, use_masks=False)
    assert torch.allclose(J1, J2, rtol=2e-5, atol=2e-6) and torch.allclose(s1, s2, rtol=0, atol=1e-6)
    assert torch.equal(s1, sdf.view(-1)[rows.long()])


def test_mlp_forward_empty(dec):
    sdf, _ = dec(torch.zeros((1, 6), device=DEV))
    assert sdf.shape == (1, 1)


# ---- band selection / surface ---------------------------------------------------------------------------------------

@pytest.mark.parametrize("G,frac", [(1, 1.0), (255, 0.3), (256, 0.0), (257, 1.0), (4096, 0.05), (64000, 0.04)])
def test_band_select_matches_nonzero(G, frac):
    from sdflabel_amd.grid import band_select
    rng = np.random.default_rng(G)
    sdf = rng.uniform(0.031, 1.0, G).astype(np.float32) * rng.choice([-1, 1], G)
    k = int(round(frac * G))
    pick = rng.choice(G, k, replace=False)
    sdf[pick] = rng.uniform(-0.0299, 0.0299, k)
    sdf = sdf.astype(np.float32)
    idx, n, slot = band_select(T(sdf), 0.03)
    ref = np.nonzero(np.abs(sdf) < np.float32(0.03))[0]
    assert n == len(ref)
    assert np.array_equal(N(idx)[:n], ref)
    s = N(slot)[:G]
    assert np.array_equal(np.nonzero(s >= 0)[0], ref) and np.array_equal(s[ref], np.arange(n))


def test_band_select_capacity_overflow_reports_true_count():
    L = _lib.lib()
    G, cap = 1000, 10
    sdf = torch.zeros(G, device=DEV)
    idx = torch.full((cap,), -7, dtype=torch.int32, device=DEV)
    cnt = torch.zeros(1, dtype=torch.int32, device=DEV)
    scratch = torch.zeros(8, dtype=torch.int32, device=DEV)
    _lib.check(L.sdfr_band_select(_lib.ptr(sdf), G, 1, 0.03, _lib.ptr(idx), cap, _lib.ptr(cnt), None, _lib.ptr(scratch),
                                  _lib.stream_ptr()), "band")
    assert int(cnt.item()) == G and np.array_equal(N(idx), np.arange(cap))


@pytest.mark.parametrize("tag", ["a", "b"])
def test_surface_points_golden(dec, tag):
    z = gold("g3_surface.npz")
    D = int(z[tag + "_D"])
    grid = sdflabel_amd.Grid3D(D, DEV)
    lat = F.normalize(T(z[tag + "_latent"]), p=2, dim=0)
    inputs = torch.cat([lat.expand(grid.points.size(0), -1), grid.points], 1)
    sdf, _ = dec(inputs)
    assert np.abs(N(sdf) - z[tag + "_sdf"]).max() < 5e-6
    pts, nocs, nrm = grid.get_surface_points(sdf)
    assert pts.shape == z[tag + "_points"].shape          # identical band (margin recorded in the golden)
    assert np.array_equal(N(sdf._sdfr_state.idx)[:pts.shape[0]], z[tag + "_band_idx"])
    assert np.abs(N(pts) - z[tag + "_points"]).max() < 1e-5
    assert np.abs(N(nocs) - z[tag + "_nocs"]).max() < 1e-5
    assert np.abs(N(nrm) - z[tag + "_normals"]).max() < 5e-5


def test_surface_points_generic_sdf_matches_fused(dec):
    """An SDF that is not the HIP decoder goes through torch.autograd.grad for its normals; an analytic sphere is exact."""
    grid = sdflabel_amd.Grid3D(12, DEV)
    sdf = (grid.points.norm(dim=1, keepdim=True) - 0.7)
    pts, nocs, nrm = grid.get_surface_points(sdf, 0.05)
    assert pts.shape[0] > 0
    assert np.abs(N(pts.norm(dim=1)) - 0.7).max() < 1e-5
    loss = (pts * pts).sum()
    loss.backward()                                         # flows to grid.points through sdf and the identity term
    assert grid.points.grad is not None and torch.isfinite(grid.points.grad).all()


def test_surface_empty_band(dec):
    grid = sdflabel_amd.Grid3D(6, DEV)
    sdf = grid.points.sum(dim=1, keepdim=True) * 0 + 0.5
    pts, nocs, nrm = grid.get_surface_points(sdf)
    assert pts.shape == (0, 3) and nocs.shape == (0, 3) and nrm.shape == (0, 3)
    r = sdflabel_amd.Rasterer(T(K_for(16, 16)), (16, 16)).to(DEV)
    rendering, points = r(pts, nrm, nrm, torch.eye(4, device=DEV), rot="dcm", output_mask=True, output_depth=True,
                          output_normals=True, output_nocs=True)
    assert float(rendering["color"].abs().max()) == 0 and float(rendering["mask"].max()) == 0
    assert points["xyzf"].shape == (0, 3)


# ---- projection / rasterer ------------------------------------------------------------------------------------------

def test_project_and_rasterer_golden():
    z = gold("g6_rasterer.npz")
    pts, nrm, col, pose = T(z["points"]), T(z["normals"]), T(z["colors"]), T(z["pose"])
    for (H, W) in ((32, 32), (64, 48)):
        t0 = "r%dx%d_" % (H, W)
        r = sdflabel_amd.Rasterer(T(z[t0 + "K"]), (W, H)).to(DEV)
        assert np.array_equal(N(r.Kinv), z[t0 + "Kinv"]) or np.allclose(N(r.Kinv), z[t0 + "Kinv"], rtol=1e-6)
        for flag, name in ((True, "nocs_"), (False, "col_")):
            _, _, _, aux = O.rasterer_forward(z[t0 + "K"], z[t0 + "Kinv"], (W, H), z["points"], z["normals"], z["colors"], z["pose"],
                                              rot="dcm", output_nocs=flag, want_aux=True)
            rend, points = r(pts, nrm, col, pose, rot="dcm", primitives="disc", bg=None, output_mask=True, output_depth=True,
                             output_normals=True, output_nocs=flag, output_points=True)
            t = t0 + name
            for k in ("color", "mask", "depth", "normals"):
                images_close(N(rend[k]), z[t + k], aux)
            for k in ("xyz", "rgb", "xyzf", "rgbf"):
                assert points[k].shape == z[t + "pts_" + k].shape, k
                assert np.abs(N(points[k]) - z[t + "pts_" + k]).max() < 1e-5, k


def test_project_golden_poses():
    z = gold("g4_project.npz")
    L = _lib.lib()
    n = z["points"].shape[0]
    pts, nrm = T(z["points"]), T(z["normals"])
    K = T(z["K"])
    for i in range(3):
        for mode, name in ((1, "nocs"), (0, "col")):
            t = "dcm%d_%s_" % (i, name)
            pose = T(z[t + "pose"])
            p_cam = torch.empty((n, 3), device=DEV); n_cam = torch.empty((n, 3), device=DEV); col = torch.empty((n, 3), device=DEV)
            uv = torch.empty((n, 2), device=DEV)
            fidx = torch.empty((n,), dtype=torch.int32, device=DEV); fcnt = torch.zeros(1, dtype=torch.int32, device=DEV)
            xyzf = torch.zeros((n, 3), device=DEV); fslot = torch.full((n,), -7, dtype=torch.int32, device=DEV)
            _lib.check(L.sdfr_project_dcm(_lib.ptr(pose), _lib.ptr(K), _lib.ptr(pts), _lib.ptr(nrm), _lib.ptr(nrm), 1, n, None, mode,
                                          32, 32, _lib.ptr(p_cam), _lib.ptr(n_cam), _lib.ptr(col), _lib.ptr(uv), _lib.ptr(fidx),
                                          _lib.ptr(fcnt), _lib.ptr(xyzf), _lib.ptr(fslot), _lib.stream_ptr()), "project")
            nf = int(fcnt[0]); fi = fidx[:nf].long()
            assert torch.equal(xyzf[:nf], p_cam[fi])                                   # fused points['xyzf'] gather
            inv = torch.full((n,), -1, dtype=torch.int32, device=DEV); inv[fi] = torch.arange(nf, dtype=torch.int32, device=DEV)
            assert torch.equal(fslot, inv)
            # modes 5 / 6: the same colours with the compositing map (c+1)/2 applied in the kernel
            if mode:
                col2 = torch.empty_like(col)
                _lib.check(L.sdfr_project_dcm(_lib.ptr(pose), _lib.ptr(K), _lib.ptr(pts), _lib.ptr(nrm), _lib.ptr(nrm), 1, n, None, mode | 4,
                                              32, 32, _lib.ptr(p_cam), _lib.ptr(n_cam), _lib.ptr(col2), None, None, None, None, None,
                                              _lib.stream_ptr()), "project")
                assert torch.equal(col2, (col + 1) / 2)
            assert np.abs(N(p_cam) - z[t + "points_3d"]).max() < 1e-5
            assert np.abs(N(n_cam) - z[t + "normals_3d"]).max() < 1e-5
            assert np.abs(N(col) - z[t + "colors_3d"]).max() < 1e-6
            assert np.abs(N(uv) - z[t + "points_2d"]).max() < 2e-4      # pixel units
            nf = int(fcnt.item())
            assert nf == z[t + "points_3d_filt"].shape[0]
            assert np.abs(N(p_cam)[N(fidx)[:nf]] - z[t + "points_3d_filt"]).max() < 1e-5


def test_rasterer_quat_mode_golden():
    z = gold("g4_project.npz")
    r = sdflabel_amd.Rasterer(T(z["K"]), (32, 32)).to(DEV)
    cam = T(z["quat_pose"])
    rend = r(T(z["points"]), T(z["normals"]), T(z["normals"]), cam, rot="quat", output_nocs=True, output_points=False,
             output_mask=True)
    Kinv = np.linalg.inv(z["K"].astype(np.float32))
    W = O.inside_surfel(Kinv, O.pixel_grid((32, 32)), z["quat_points_3d"], z["quat_normals_3d"], diam=0.04)
    ref = np.minimum((W.T @ ((z["quat_colors_3d"] + 1) / 2)).T, 1).reshape(3, 32, 32)
    images_close(N(rend["color"]), ref)
    with pytest.raises(KeyError):
        r(T(z["points"]), T(z["normals"]), T(z["normals"]), cam, rot="quat", output_nocs=True)


# ---- splat kernels vs oracle on adversarial surfels ------------------------------------------------------------------

def _random_surfels(rng, n, H, W):
    p = np.stack([rng.uniform(-0.6, 0.6, n), rng.uniform(-0.6, 0.6, n), rng.uniform(0.8, 2.0, n)], 1).astype(np.float32)
    nrm = rng.standard_normal((n, 3)).astype(np.float32) * 0.5 + np.array([0, 0, -1], np.float32)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    # grazing normals (|n.ray| < 0.01 for some pixels), a surfel behind the camera, one far off screen, one near the camera plane
    nrm[:3] = np.array([[1, 0, 0.001], [0, 1, -0.002], [0.7071, 0.7071, 0.0]], np.float32)
    p[3] = [0.1, 0.1, -1.5]
    p[4] = [30.0, 0.0, 1.0]
    p[5] = [0.01, -0.02, 0.03]
    col = rng.uniform(0, 1.4, (n, 3)).astype(np.float32)       # > 1 exercises the clamp(max=1) gates
    return p, nrm, col


@pytest.mark.parametrize("kshift", [(0.0, 0.0), (-23.5, 6.25)])       # principal point at the centre / well outside the image (cropped intrinsics)
@pytest.mark.parametrize("H,W,n", [(16, 16, 40), (40, 24, 300), (17, 31, 129), (24, 24, 1500), (8, 8, 1100)])   # the last two: > 256 and > 1024 candidates per tile
def test_splat_forward_backward_vs_oracle(H, W, n, kshift):
    rng = np.random.default_rng(H * 100 + n)
    p, nrm, col = _random_surfels(rng, n, H, W)
    K = K_for(H, W)
    if kshift[0] or kshift[1]:
        # the principal point moves out of the image, the surfels with it (so that they stay in view), and the focal lengths differ
        K[0, 2] += kshift[0]; K[1, 2] += kshift[1]; K[1, 1] *= 0.93
        p[6:, 0] += kshift[0] / K[0, 0] * p[6:, 2]
        p[6:, 1] += kshift[1] / K[1, 1] * p[6:, 2]
    Kinv = np.linalg.inv(K).astype(np.float32)
    L = _lib.lib()
    tp, tn, tc = T(p), T(nrm), T(col)
    color = torch.empty((3, H, W), device=DEV); mask = torch.empty((1, H, W), device=DEV)
    depth = torch.empty((1, H, W), device=DEV); nimg = torch.empty((3, H, W), device=DEV)
    aux = torch.empty((H * W, 4), device=DEV); bbox = torch.empty((n, 4), dtype=torch.int32, device=DEV)
    tK, tKi = T(K), T(Kinv)
    _lib.check(L.sdfr_splat_forward(0, _lib.ptr(tK), _lib.ptr(tKi), _lib.ptr(tp), _lib.ptr(tn), _lib.ptr(tc), None, None, None, None, 1, n, None, W, H, 0.04, 150.0,
                                    _lib.ptr(bbox), _lib.ptr(color), _lib.ptr(mask), _lib.ptr(depth), _lib.ptr(nimg), _lib.ptr(aux),
                                    _lib.stream_ptr()), "splat_fwd")
    Wm, auxo = O.inside_surfel(Kinv, O.pixel_grid((W, H)), p, nrm, diam=0.04, want_aux=True)
    ref_c = np.minimum((Wm.T @ col).T, 1).reshape(3, H, W)
    ref_m = np.minimum(Wm.sum(0), 1).reshape(1, H, W)
    ref_d = (Wm.T @ p[:, 2]).reshape(1, H, W)
    ref_n = np.minimum((Wm.T @ ((nrm + 1) / 2)).T, 1).reshape(3, H, W)
    images_close(N(color), ref_c, auxo); images_close(N(mask), ref_m, auxo)
    images_close(N(depth), ref_d, auxo); images_close(N(nimg), ref_n, auxo)
    # backward against the oracle's restatement of autograd
    gC = rng.standard_normal((3, H, W)).astype(np.float32); gM = rng.standard_normal((1, H, W)).astype(np.float32)
    gD = rng.standard_normal((1, H, W)).astype(np.float32); gN = rng.standard_normal((3, H, W)).astype(np.float32)
    g_p = torch.zeros((n, 3), device=DEV); g_n = torch.zeros((n, 3), device=DEV); g_a = torch.zeros((n, 3), device=DEV)
    tg = [T(g) for g in (gC, gM, gD, gN)]
    _lib.check(L.sdfr_splat_backward(0, _lib.ptr(tK), _lib.ptr(tKi), _lib.ptr(tp), _lib.ptr(tn), _lib.ptr(tc), None, None, None, None, 1, n, None, W, H, 0.04,
                                     150.0, _lib.ptr(aux), _lib.ptr(color), _lib.ptr(mask), _lib.ptr(depth), _lib.ptr(nimg),
                                     _lib.ptr(tg[0]), _lib.ptr(tg[1]), _lib.ptr(tg[2]), _lib.ptr(tg[3]), _lib.ptr(g_p), _lib.ptr(g_n),
                                     _lib.ptr(g_a), _lib.stream_ptr()), "splat_bwd")
    r_p, r_n, r_c = O.splat_backward(Kinv, (W, H), p, nrm, col, gC, gM, gD, gN)
    for got, ref in ((g_p, r_p), (g_n, r_n), (g_a, r_c)):
        scale = max(1.0, np.abs(ref).max())
        assert np.abs(N(got) - ref).max() < 2e-3 * scale, (np.abs(N(got) - ref).max(), scale)


def test_splat_backward_is_linear_and_deterministic():
    """size-independent properties at the full 256x256 crop: backward is linear in the upstream gradient, bitwise repeatable."""
    H = W = 256
    rng = np.random.default_rng(9)
    n = 2500
    p, nrm, col = _random_surfels(rng, n, H, W)
    p[:, :2] *= 0.5
    p[:, 2] = rng.uniform(3.0, 4.0, n)
    r = sdflabel_amd.Rasterer(T(K_for(H, W)), (W, H)).to(DEV)
    pose = torch.eye(4, device=DEV)

    def grads(wc, wm):
        tp = T(p).requires_grad_(True)
        tn = T(nrm).requires_grad_(True)
        rend = r(tp, tn, T(col), pose, rot="dcm", output_mask=True, output_depth=True, output_normals=True, output_nocs=False,
                 output_points=False)
        ((rend["color"] * wc).sum() + (rend["depth"] * wm).sum()).backward()
        return tp.grad.clone(), tn.grad.clone(), rend

    wc = torch.randn(3, H, W, device=DEV); wm = torch.randn(1, H, W, device=DEV)
    a = grads(wc, wm); b = grads(wc, wm); c = grads(2 * wc, 2 * wm)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert torch.allclose(c[0], 2 * a[0], rtol=1e-5, atol=1e-6) and torch.allclose(c[1], 2 * a[1], rtol=1e-5, atol=1e-6)
    m = a[2]["mask"]
    assert set(np.unique(N(m)).tolist()) <= {0.0, 1.0}
    assert float(m.sum()) > 500


# ---- end to end through the drop-in boundary ---------------------------------------------------------------------------

@pytest.mark.parametrize("tag", ["a", "b"])
def test_end_to_end_gradients_golden(dec, tag):
    """The optimizer's graph (optimizer.py:79-123) on top of the drop-in modules vs. reference autograd (golden G7)."""
    z = gold("g7_grads.npz")
    D, H, W = [int(v) for v in z[tag + "_cfg"]]
    grid = sdflabel_amd.Grid3D(D, DEV)
    lat = T(z[tag + "_latent"]).requires_grad_(True)
    yaw = T(z[tag + "_yaw"]).requires_grad_(True)
    trans = T(z[tag + "_trans"]).requires_grad_(True)
    renderer = sdflabel_amd.Rasterer(T(z[tag + "_K"]), (W, H)).to(DEV)
    lat_ = F.normalize(lat, p=2, dim=0)
    inputs = torch.cat([lat_.expand(grid.points.size(0), -1), grid.points], 1)
    sdf, _ = dec(inputs)
    pcd, _, normals = grid.get_surface_points(sdf)
    assert np.abs(N(pcd) - z[tag + "_pcd"]).max() < 1e-5
    pcd.retain_grad()
    pose = build_pose(yaw, trans)
    rendering, points = renderer(pcd, normals, normals, pose, primitives="disc", rot="dcm", bg=None, output_depth=True,
                                 output_normals=True, output_nocs=True, output_points=True, output_mask=True)
    for k in ("color", "mask", "depth", "normals"):
        images_close(N(rendering[k]), z[tag + "_out_" + k])
    loss = sum((rendering[k] * T(z[tag + "_W_" + k])).sum() for k in ("color", "mask", "depth", "normals"))
    loss = loss + sum((points[k] * T(z[tag + "_Wp_" + k])).sum() for k in ("xyzf", "rgbf", "xyz", "rgb"))
    assert abs(float(loss) - float(z[tag + "_loss"])) < 2e-3 * max(1.0, abs(float(z[tag + "_loss"])))
    loss.backward()
    for got, key in ((pcd.grad, "_g_pcd"), (yaw.grad, "_g_yaw"), (trans.grad, "_g_trans"), (lat.grad, "_g_latent")):
        ref = z[tag + key]
        assert np.abs(N(got) - ref).max() < 1e-3 * max(1.0, np.abs(ref).max()), key


def test_full_size_crop_vs_oracle_sample(dec, oracle_layers):
    """D = 40 grid, 128x128 crop: the rendered images against the oracle on the same surfels (the 256x256 size of BASELINE configs[1] is
    tested against reference-held data in tests/test_gpu_configs.py)."""
    layers, spec = oracle_layers
    D, H, W = 40, 128, 128
    grid = sdflabel_amd.Grid3D(D, DEV)
    lat = F.normalize(torch.tensor([0.3, -0.5, 0.8], device=DEV), p=2, dim=0)
    inputs = torch.cat([lat.expand(grid.points.size(0), -1), grid.points], 1)
    sdf, _ = dec(inputs)
    ref_sdf = O.decoder_forward(layers, spec, N(inputs))
    assert np.abs(N(sdf) - ref_sdf).max() < 1e-5
    pcd, _, normals = grid.get_surface_points(sdf)
    assert 1500 < pcd.shape[0] < 5000
    K = K_for(H, W)
    pose = build_pose(torch.tensor([0.6], device=DEV), torch.tensor([0.0, 0.0, 3.5], device=DEV))
    r = sdflabel_amd.Rasterer(T(K), (W, H)).to(DEV)
    rend, points = r(pcd, normals, normals, pose, rot="dcm", output_mask=True, output_depth=True, output_normals=True,
                     output_nocs=True)
    ro, po, _, aux = O.rasterer_forward(K, np.linalg.inv(K).astype(np.float32), (W, H), N(pcd), N(normals), N(normals), N(pose),
                                        rot="dcm", output_nocs=True, want_aux=True)
    for k in ("color", "mask", "depth", "normals"):
        images_close(N(rend[k]), ro[k], aux)
    assert np.abs(N(points["xyzf"]) - po["xyzf"]).max() < 1e-5


G8_FILES = ["g8_optimizer.npz", "g8b_optimizer_128.npz"]       # 32x32 / D=20, and BASELINE configs[0]'s size: 128x128 / D=40


@pytest.mark.parametrize("gfile", G8_FILES)
def test_refinement_trajectory_golden(dec, gfile):
    """a8 / a-harness: 10 iterations of the reference's refinement loop (Adam + SGD, 2-D NOCS window loss, 3-D NN loss) restated in
    tests/_harness.py on top of the drop-in modules, against the trajectory the reference's own Optimizer produced (goldens G8, G8b)."""
    from tests._harness import Refiner
    z = gold(gfile)
    D, H, W = int(z["D"]), int(z["H"]), int(z["W"])
    init = z["init"]
    params = {"yaw": init[0:1], "trans": init[1:4], "scale": init[4:5], "latent": init[5:8]}
    ref = Refiner(params, DEV, {"2d": 0.3, "3d": 0.5})
    grid = sdflabel_amd.Grid3D(D, DEV)
    renderer = sdflabel_amd.Rasterer(T(z["K"]), (W, H)).to(DEV)
    traj = []
    for _ in range(10):
        ref.optimize(1, T(z["nocs_target"]), z["lidar"], dec, grid, renderer)
        traj.append(ref.vector())
    traj = np.asarray(traj)
    assert len(ref.log) == 10
    l2 = np.array([a for a, _ in ref.log]); l3 = np.array([b for _, b in ref.log])
    assert np.abs(l2 - z["loss2d_weighted"]).max() < 2e-4, (l2, z["loss2d_weighted"])
    assert np.abs(l3 - z["loss3d_weighted"]).max() < 2e-4
    assert np.abs(traj - z["traj"]).max() < 5e-4, np.abs(traj - z["traj"]).max(axis=0)
    assert abs(traj[-1, 0] - init[0]) > 0.05            # the pose really moved


@pytest.mark.parametrize("prim,use_bg", [("circle", False), ("circle", True), ("circle_opt", False), ("circle_opt", True), ("disc", True)])
def test_secondary_primitives_and_bg_golden(prim, use_bg):
    """a6' rows: primitives 'circle' / 'circle_opt' and the bg variant through the drop-in Rasterer vs. the reference (golden G9):
    images and autograd gradients w.r.t. the surfel positions, yaw and trans."""
    z = gold("g9_secondary.npz")
    H = W = 32
    t = "%s_bg%d_" % (prim, int(use_bg))
    r = sdflabel_amd.Rasterer(T(z["K"]), (W, H)).to(DEV)
    p = T(z["points"]).requires_grad_(True)
    nrm = T(z["normals"])
    yaw = torch.tensor([0.6], device=DEV, requires_grad=True)
    trans = torch.tensor([0.05, -0.03, 3.4], device=DEV, requires_grad=True)
    pose = build_pose(yaw, trans)
    assert np.allclose(N(pose), z[t + "pose"], atol=1e-6)
    rend = r(p, nrm, nrm, pose, rot="dcm", primitives=prim, bg=T(z["bg"]) if use_bg else None, output_mask=True,
             output_depth=not use_bg, output_normals=not use_bg, output_nocs=True, output_points=False)
    # circle_opt scales the normalised depth by 10000 before the softmax: one float32 ulp of the logit moves the weights by ~2e-4
    tol = 1e-3 if prim == "circle_opt" else 1e-4
    for k in rend:
        assert np.abs(N(rend[k]) - z[t + "out_" + k]).max() < tol, k
    loss = sum((rend[k] * T(z[t + "W_" + k])).sum() for k in rend)
    loss.backward()
    gtol = 2e-2 if prim == "circle_opt" else 2e-3
    for got, key in ((p.grad, "g_points"), (yaw.grad, "g_yaw"), (trans.grad, "g_trans")):
        ref = z[t + key]
        assert np.abs(N(got) - ref).max() < gtol * max(1.0, np.abs(ref).max()), (key, np.abs(N(got) - ref).max(), np.abs(ref).max())


def test_bg_with_depth_or_normals_is_rejected_like_the_reference():
    r = sdflabel_amd.Rasterer(T(K_for(16, 16)), (16, 16)).to(DEV)
    p = torch.rand(5, 3, device=DEV) + torch.tensor([0, 0, 2.0], device=DEV)
    with pytest.raises(RuntimeError):
        r(p, p, p, torch.eye(4, device=DEV), rot="dcm", bg=torch.zeros(3, 16, 16, device=DEV), output_depth=True, output_points=False)


def test_splat_candidate_list_overflow_path_vs_oracle():
    """more than SPL_LC = 1024 surfels overlapping one 8x8 tile: the tile falls back to walking every surfel; same result."""
    rng = np.random.default_rng(21)
    H = W = 16
    n = 1500
    p = np.stack([rng.uniform(-0.05, 0.05, n), rng.uniform(-0.05, 0.05, n), rng.uniform(1.0, 1.2, n)], 1).astype(np.float32)
    nrm = rng.standard_normal((n, 3)).astype(np.float32) * 0.3 + np.array([0, 0, -1], np.float32)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    col = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    K = K_for(H, W)
    Kinv = np.linalg.inv(K).astype(np.float32)
    r = sdflabel_amd.Rasterer(T(K), (W, H)).to(DEV)
    tp = T(p).requires_grad_(True)
    rend = r(tp, T(nrm), T(col), torch.eye(4, device=DEV), rot="dcm", output_mask=True, output_depth=True, output_normals=True,
             output_nocs=False, output_points=False)
    Wm, aux = O.inside_surfel(Kinv, O.pixel_grid((W, H)), p, nrm, diam=0.04, want_aux=True)
    assert (Wm > 0).sum(axis=0).max() > 200                     # hundreds of surfels composite into single pixels
    images_close(N(rend["color"]), np.minimum((Wm.T @ col).T, 1).reshape(3, H, W), aux)
    images_close(N(rend["depth"]), (Wm.T @ p[:, 2]).reshape(1, H, W), aux)
    gC = rng.standard_normal((3, H, W)).astype(np.float32)
    (rend["color"] * T(gC)).sum().backward()
    r_p, _, _ = O.splat_backward(Kinv, (W, H), p, nrm, col, gC, None, None, None)
    assert np.abs(N(tp.grad) - r_p).max() < 2e-3 * max(1.0, np.abs(r_p).max())


def test_full_band_stress_all_grid_points_are_surfels():
    """a random-initialised decoder puts (nearly) every grid point inside the band: N ~ G surfels through Jacobian, projection and
    splat; checked against the oracle on the decoder outputs and through size-independent properties on the images."""
    torch.manual_seed(3)
    d = sdflabel_amd.Decoder(3, dims=[64] * 4, norm_layers=(), latent_in=[2], weight_norm=False).to(DEV).eval()
    grid = sdflabel_amd.Grid3D(12, DEV)
    G = grid.points.shape[0]
    lat = torch.tensor([0.1, 0.2, -0.3], device=DEV)
    inputs = torch.cat([lat.expand(G, -1), grid.points], 1)
    sdf, _ = d(inputs)
    pts, nocs, nrm = grid.get_surface_points(sdf, threshold=10.0)
    assert pts.shape[0] == G
    layers = [(W, b, None) for W, b in d.effective_layers()]
    spec = dict(dims=[64] * 4, latent_in=[2])
    ref, cache = O.decoder_forward(layers, spec, N(inputs), want_cache=True)
    J = O.decoder_backward_inputs(layers, spec, N(inputs), cache, np.ones_like(ref))
    pm, _, nm, _, _ = O.get_surface_points(N(grid.points), ref, J[:, 3:], 10.0)
    assert np.abs(N(sdf) - ref).max() < 5e-6 and np.abs(N(pts) - pm).max() < 5e-5
    r = sdflabel_amd.Rasterer(T(K_for(48, 48)), (48, 48)).to(DEV)
    pose = build_pose(torch.tensor([0.3], device=DEV), torch.tensor([0.0, 0.0, 3.0], device=DEV))
    rend, points = r(pts, nrm, nrm, pose, rot="dcm", output_mask=True, output_depth=True, output_nocs=True)
    m = N(rend["mask"])
    assert set(np.unique(m).tolist()) <= {0.0, 1.0} and m.sum() > 100
    dep = N(rend["depth"])
    assert (dep[m > 0] > 1.0).all() and (dep[m == 0] == 0).all()
    assert N(rend["color"]).max() <= 1            # clamp(max=1); NOCS of points projected from far outside the cube may be negative
    back = ((N(nrm) @ N(pose)[:3, :3].T) * N(points["xyz"])).sum(1) >= 0
    assert points["xyzf"].shape[0] + int(back.sum()) == G        # front-facing filter partitions the surfels


def test_512_crop_batch_equals_dropin(dec):
    """BASELINE configs[4] image size: the batched path and the drop-in path agree at 512x512, D = 40."""
    from tests.test_gpu_batch import dropin_step
    D, H, W = 40, 512, 512
    K = K_for(H, W)
    br = sdflabel_amd.BatchRenderer(dec, D, K, (W, H), 1, device=DEV)
    yaw, trans, lat = np.array([0.6], np.float32), np.array([[0.0, 0.0, 3.5]], np.float32), np.array([[0.3, -0.5, 0.8]], np.float32)
    out = br.forward(T(yaw), T(trans), T(lat))
    w = {"color": torch.randn(1, 3, H, W, device=DEV), "mask": torch.randn(1, 1, H, W, device=DEV),
         "depth": torch.randn(1, 1, H, W, device=DEV), "normals": torch.randn(1, 3, H, W, device=DEV),
         "xyzf": torch.randn(1, br.cap, 3, device=DEV)}
    g = br.backward(g_color=w["color"], g_mask=w["mask"], g_depth=w["depth"], g_normals=w["normals"], g_xyzf=w["xyzf"])
    rend, pts, n, grads = dropin_step(dec, D, H, W, K, yaw[0], trans[0], lat[0], {k: v[0] for k, v in w.items()})
    assert int(out["n"][0]) == n and float(out["mask"].sum()) > 20000
    for k in ("color", "mask", "depth", "normals"):
        assert np.abs(N(out[k][0]) - N(rend[k])).max() < 2e-5, k
    for got, ref in ((g[0], grads[0]), (g[1][0], grads[1]), (g[2][0], grads[2])):
        ref = N(ref)
        assert np.abs(N(got) - ref).max() < 5e-4 * max(1.0, np.abs(ref).max())


# ---- float16 decoder (the reference's default precision, configs/config_refine.ini:19; BASELINE configs[4]) ------------------------

def _half(a):
    return a.astype(np.float16).astype(np.float32)


def _emulate_f16_decoder(layers, spec, inp):
    """numpy model of sdfr_mlp_forward_f16: weights and hidden activations rounded to half, float32 accumulation, bias, ReLU and the
    last 512->1 dot in float32.  Returns sdf, and the Jacobian the float32 mask-fed backward must produce (float32 weights, the
    ReLU masks of THIS forward)."""
    n_lin = len(layers)
    x = _half(inp)
    masks = []
    for l in range(n_lin - 1):
        W, b, _ = layers[l]
        if l in spec["latent_in"]:
            x = np.concatenate([x, _half(inp)], 1)
        y = x @ _half(W).T + b
        masks.append(y > 0)
        x = _half(np.maximum(y, 0))
    W, b, _ = layers[-1]
    o = np.tanh(x @ W.T + b)
    g = (1 - o * o)                                   # (n,1)
    g = g * W                                         # d/d act7  (n,512)
    J = np.zeros_like(inp)
    for l in range(n_lin - 2, -1, -1):
        g = g * masks[l]
        g = g @ layers[l][0]
        if l in spec["latent_in"]:
            J += g[:, -inp.shape[1]:]
            g = g[:, :-inp.shape[1]]
    J += g
    return o.astype(np.float32), J.astype(np.float32)


def test_f16_decoder_vs_half_rounding_model(oracle_layers):
    layers, spec = oracle_layers
    dec16, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=torch.float16)
    dec16 = dec16.to(DEV)
    rng = np.random.default_rng(16)
    for n in (1, 127, 128, 129, 1000):
        inp = (rng.standard_normal((n, 6)) * 0.6).astype(np.float32)
        sdf, _ = dec16(T(inp))
        ref, Jref = _emulate_f16_decoder(layers, spec, inp)
        assert np.abs(N(sdf) - ref).max() < 2e-4, n              # same rounding points; only the f32 summation order differs
    # and against the float32 decoder: half operands cost ~1e-3
    ref32 = O.decoder_forward(layers, spec, inp)
    assert 1e-6 < np.abs(N(sdf) - ref32).max() < 1e-2
    # mask-fed float32 Jacobian on top of the f16 forward
    from sdflabel_amd.deepsdf.networks.deep_sdf_decoder_scale import mlp_jacobian
    rows = T(np.sort(rng.choice(n, 300, replace=False)).astype(np.int32))
    J, sel = mlp_jacobian(sdf._sdfr_state, rows, 300, half=False)
    assert torch.equal(sel, sdf.view(-1)[rows.long()])
    Jr = Jref[N(rows)]
    # the default of the float16 decoder: the backward runs with half operands as well (weights and in-gradients rounded to half, float32
    # accumulation) -- relative 1e-3 class, like the forward
    Jh, selh = mlp_jacobian(sdf._sdfr_state, rows, 300)
    assert torch.equal(selh, sel)
    errh = np.abs(N(Jh) - Jr)
    assert np.median(errh) < 1e-3 and np.quantile(errh, 0.98) < 1e-2 and errh.max() < 5e-2, (np.median(errh), errh.max())
    # a hidden unit whose pre-activation is within float rounding of 0 may get the other ReLU mask bit in the model (4 M units here):
    # a handful of rows differ at the 1e-3 level, everything else agrees to rounding
    err = np.abs(N(J) - Jr)
    assert np.median(err) < 1e-5 and np.quantile(err, 0.98) < 5e-4 and err.max() < 2e-2


def test_f16_decoder_end_to_end_close_to_f32(dec):
    """the whole crop-iteration with the float16 decoder: band and images stay close to the float32 path; the differing pixels are the
    silhouette / band-boundary ones (stated tolerance for BASELINE configs[4]: < 3 % of the pixels differ by more than 1e-2)."""
    dec16, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=torch.float16)
    dec16 = dec16.to(DEV)
    D, H, W = 40, 128, 128
    K = K_for(H, W)
    yaw, trans, lat = T(np.array([0.6], np.float32)), T(np.array([[0.0, 0.0, 3.5]], np.float32)), T(np.array([[0.3, -0.5, 0.8]], np.float32))
    outs = []
    for d in (dec, dec16):
        br = sdflabel_amd.BatchRenderer(d, D, K, (W, H), 1, device=DEV)
        o = br.forward(yaw, trans, lat)
        g = br.backward(g_color=torch.ones(1, 3, H, W, device=DEV), g_xyzf=torch.ones(1, br.cap, 3, device=DEV))
        outs.append((int(o["n"][0]), N(o["color"][0]), N(o["mask"][0]), N(br.sdf), [N(t).copy() for t in g]))
    n32, c32, m32, s32, g32 = outs[0]
    n16, c16, m16, s16, g16 = outs[1]
    assert np.abs(s16 - s32).max() < 1e-2 and np.abs(s16 - s32).mean() < 1e-3
    assert abs(n16 - n32) < 0.1 * n32
    assert (np.abs(c16 - c32).max(axis=0) > 1e-2).mean() < 0.03
    assert (m16 != m32).mean() < 0.01
    assert all(np.isfinite(g).all() for g in g16)
    assert abs(g16[0][0] - g32[0][0]) < 0.25 * max(1.0, abs(g32[0][0]))


def test_half_precision_callers_are_served(dec):
    """the reference's default config runs everything in float16 (configs/config_refine.ini:19): Grid3D(precision=half), half inputs,
    half K and pose.  The drop-in widens at the boundary, computes in float32 (decoder MFMAs in half) and narrows the results back;
    the rendering stays close to the float32 path and gradients reach the float32 leaf parameters."""
    dec16, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision=torch.float16)
    dec16 = dec16.to(DEV)
    D, H, W = 30, 64, 64
    K = T(K_for(H, W))
    res = {}
    for prec, d in ((torch.float32, dec), (torch.float16, dec16)):
        grid = sdflabel_amd.Grid3D(D, DEV, prec)
        assert grid.points.dtype == prec
        lat = torch.tensor([0.3, -0.5, 0.8], device=DEV, requires_grad=True)
        yaw = torch.tensor([0.6], device=DEV, requires_grad=True)
        trans = torch.tensor([0.0, 0.0, 3.5], device=DEV, requires_grad=True)
        renderer = sdflabel_amd.Rasterer(K.to(prec), (W, H), precision=prec).to(DEV)
        lat_ = F.normalize(lat.to(prec), p=2, dim=0)
        inputs = torch.cat([lat_.expand(grid.points.size(0), -1), grid.points], 1).to(lat_.device, lat_.dtype)
        sdf, scale = d(inputs)
        assert sdf.dtype == prec and scale.dtype == prec
        pcd, nocs, nrm = grid.get_surface_points(sdf)
        assert pcd.dtype == prec and nrm.dtype == prec
        pose = build_pose(yaw, trans).to(prec)
        rend, pts = renderer(pcd, nrm, nrm, pose, primitives='disc', rot='dcm', output_normals=True, output_nocs=True, output_mask=True)
        assert rend["color"].dtype == prec and pts["xyzf"].dtype == prec
        loss = rend["color"].float().sum() + pts["xyzf"].float().sum()
        loss.backward()
        res[prec] = (N(rend["color"].float()), N(rend["mask"].float()), pcd.shape[0], [N(g) for g in (yaw.grad, trans.grad, lat.grad)])
        assert all(np.isfinite(g).all() for g in res[prec][3])
    c32, m32, n32, g32 = res[torch.float32]
    c16, m16, n16, g16 = res[torch.float16]
    assert abs(n16 - n32) < 0.1 * n32
    assert (m16 != m32).mean() < 0.02 and (np.abs(c16 - c32).max(axis=0) > 2e-2).mean() < 0.05
    assert abs(g16[0][0] - g32[0][0]) < 0.3 * max(1.0, abs(g32[0][0]))


# ---- error-compensated f16 decoder ("float32_split"): float32 results from the f16 matrix cores -----------------------------------

@pytest.fixture(scope="module")
def dec_split():
    d, _ = sdflabel_amd.setup_dsdf(ASSET + ".pt", precision="float32_split")
    return d.to(DEV)


def test_split_decoder_is_float32_accurate(dec, dec_split, oracle_layers):
    """hi/lo half operand pairs, three f16 MFMAs per product: the output must sit as close to a float64 evaluation of the network as the
    exact-f32 kernel does, the band must be the same rows and the saved ReLU masks (same layout) may differ only in a handful of bits"""
    layers, spec = oracle_layers
    grid = sdflabel_amd.Grid3D(40, DEV)
    lat = F.normalize(torch.tensor([0.3, -0.5, 0.8], device=DEV), p=2, dim=0)
    inputs = torch.cat([lat.expand(grid.points.size(0), -1), grid.points.detach()], 1).contiguous()
    with torch.no_grad():
        s32, _ = dec(inputs)
        ssp, _ = dec_split(inputs)
    assert ssp._sdfr_state.split and not s32._sdfr_state.split
    sel = np.arange(0, inputs.shape[0], 9)
    ref = O.decoder_forward(layers, spec, N(inputs)[sel].astype(np.float64)).reshape(-1)      # the oracle evaluated in float64
    e32 = np.abs(N(s32).reshape(-1)[sel] - ref).max()
    esp = np.abs(N(ssp).reshape(-1)[sel] - ref).max()
    assert esp < 5e-7 and esp < 2.0 * e32 + 1e-7, (esp, e32)
    assert float((s32 - ssp).abs().max()) < 1e-6
    assert int(((s32.abs() < 0.03) != (ssp.abs() < 0.03)).sum()) <= 2
    m32, msp = s32._sdfr_state.mask_ws, ssp._sdfr_state.mask_ws
    assert m32.numel() == msp.numel()
    x = m32 ^ msp
    flips = int(sum(int(((x >> b) & 1).sum()) for b in range(32)))
    assert flips <= 1e-6 * 32 * m32.numel() + 64, flips


def test_split_decoder_passes_the_float32_goldens(dec_split, oracle_layers):
    """the same golden vectors and tolerances as the exact-f32 path, through the drop-in modules"""
    test_mlp_forward_golden_fitted(dec_split)
    test_mlp_backward_golden_fitted(dec_split)
    for n in (1, 63, 65, 1000):
        test_mlp_forward_ragged_sizes_vs_oracle(dec_split, oracle_layers, n)
    for tag in ("a", "b"):
        test_surface_points_golden(dec_split, tag)
        test_end_to_end_gradients_golden(dec_split, tag)
    test_full_size_crop_vs_oracle_sample(dec_split, oracle_layers)
    test_refinement_trajectory_golden(dec_split, "g8_optimizer.npz")


def test_c_abi_smoke_binary_runs_without_python_or_torch(tmp_path):
    """the boundary driven from plain C (tests/c_abi/abi_smoke.c): decoder create / forward / Jacobian / destroy against a double-precision
    host evaluation, device memory from the HIP runtime directly"""
    import subprocess
    from tests._util import build_c_abi_smoke
    exe = build_c_abi_smoke(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.stdout + r.stderr)[-1500:]
    assert "C ABI smoke: OK" in r.stdout
