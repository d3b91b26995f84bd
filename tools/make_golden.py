"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE (build container only).

The reference (TRI-ML/sdflabel @ /root/reference) ships no tests or fixtures for the renderer hot path
(SURVEY.md §4), so parity is pinned by vectors produced here from the reference's own code, imported
read-only with bytecode writing disabled (tools/_ref_import.py).  Only DATA is committed: inputs, outputs,
seeds, torch version and threshold margins -- never reference source.

  G1 grid            grid.Grid3D.generate_point_grid                       (grid.py:22-41)
  G2 decoder         Decoder.forward + d(sum sdf)/d inputs                 (deep_sdf_decoder_scale.py:78-114)
  G3 surface         Grid3D.get_surface_points                             (grid.py:43-71)
  G4 project         project_in_2D / project_in_2D_quat                    (projection.py:7-199)
  G5 inside_surfel   inside_surfel weights (+bg variant)                   (primitives.py:165-242)
  G6 rasterer        Rasterer.forward, 32x32 and 64x64, nocs T/F, bg       (rasterer.py:49-155)
  G7 grads           autograd gradients of a fixed random functional of all outputs w.r.t.
                     yaw, trans, latent, coords, normals                  (optimizer.py:79-123 graph)
  G8 optimizer       10-iteration Optimizer.optimize trajectory            (optimizer.py:56-164)
  G9 secondary       Rasterer.forward with primitives circle / circle_opt and bg, + gradients   (primitives.py:4-162)
  G10 config1        BASELINE configs[1] at full size: 256x256, D=40, float32, images + surfels + gradients        (optimizer.py:79-123 graph)
  G11 config4        the reference's own float16 run at 512x512, D=40, beside its float32 run (config_refine.ini:19)
  G13 primitives     standalone inside_surfel / inside_circle / inside_circle_opt weights + gradients, project_in_2D(_quat) gradients
  G14 cropped        Rasterer.forward + gradients with crop intrinsics from the reference's adjust_intrinsics_crop (utils/refinement.py:586-609);
                     G14o the Optimizer trajectory at rendering_area = 32 with them; G14e a second decoder (ellipsoid fit, + LayerNorm variant) at full size
  G12 losses         compute_loss_2d / compute_loss_3d values and gradients                                        (optimizer.py:166-237)

usage: python tools/make_golden.py [G1 G2 ...]
"""
import contextlib
import io
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _ref_import  # noqa: E402

_ref_import.setup()
sys.modules["pyquaternion"].Quaternion = object  # `from pyquaternion import Quaternion` (utils/refinement.py:6)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import grid as ref_grid  # noqa: E402  (reference sdfrenderer/grid.py)
from deepsdf.networks.deep_sdf_decoder_scale import Decoder  # noqa: E402
import deepsdf.workspace as ref_ws  # noqa: E402
from renderer.rasterer import Rasterer  # noqa: E402
from renderer import projection as ref_proj  # noqa: E402
from renderer import primitives as ref_prim  # noqa: E402
import utils.refinement as rtools  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "tests", "golden")
ASSET = os.path.join(HERE, "..", "sdflabel_amd", "assets", "deepsdf_synth.pt")
META = dict(torch_version=str(torch.__version__), numpy_version=str(np.__version__))
torch.set_num_threads(8)


def save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **{k: (np.asarray(v) if not isinstance(v, str) else np.asarray(v)) for k, v in arrs.items()},
                        **{"_" + k: np.asarray(v) for k, v in META.items()})
    print("wrote", name, "%.1f KB" % (os.path.getsize(path) / 1024))


def state_np(mod):
    return {k: v.detach().cpu().float().numpy() for k, v in mod.state_dict().items()}


def load_fitted(precision=torch.float32):
    dec, L = ref_ws.setup_dsdf(ASSET, precision=precision)
    return dec, L


def K_for(H, W):
    f = 45.0 * H / 32.0
    return torch.tensor([[f, 0, W / 2.0], [0, f, H / 2.0], [0, 0, 1]], dtype=torch.float32)


def build_pose(yaw, trans):
    """pipelines/optimizer.py:86-90 executed verbatim."""
    render_pose = torch.eye(4)
    render_pose[:3, :3] = rtools.rot_from_yaw(yaw)
    render_pose[1] *= -1
    render_pose[:3, 3] = trans
    return render_pose


# ---------------------------------------------------------------------------------------------------------

def g1():
    arrs = {}
    for D in (4, 5, 8, 30, 40):
        g = ref_grid.Grid3D(D, "cpu", torch.float32).points.detach().numpy()
        if D <= 8:
            arrs["grid_%d" % D] = g
        else:
            arrs["grid_%d_stride97" % D] = g[::97]
            arrs["grid_%d_sum64" % D] = g.astype(np.float64).sum(0)
            arrs["grid_%d_tail" % D] = g[-8:]
    save("g1_grid.npz", **arrs)


def g2():
    arrs = {}
    torch.manual_seed(7)
    # small seeded nets: weight-norm variant and LayerNorm variant, latent_in=[4]
    for tag, wn in (("wn", True), ("ln", False)):
        dec = Decoder(3, dims=[64] * 8, dropout=list(range(8)), dropout_prob=0.2, norm_layers=list(range(8)),
                      latent_in=[4], weight_norm=wn, xyz_in_all=False, use_tanh=False, latent_dropout=False)
        dec.eval()
        with torch.no_grad():
            for p in dec.parameters():      # de-trivialise biases / gains
                p.add_(0.05 * torch.randn_like(p))
        inp = torch.randn(300, 6) * 0.7
        inp.requires_grad_(True)
        sdf, scale = dec(inp)
        sdf.sum().backward()
        for k, v in state_np(dec).items():
            arrs["%s_state_%s" % (tag, k)] = v
        arrs["%s_inputs" % tag] = inp.detach().numpy()
        arrs["%s_sdf" % tag] = sdf.detach().numpy()
        arrs["%s_scale" % tag] = scale.detach().numpy()
        arrs["%s_grad_inputs" % tag] = inp.grad.numpy()
    # an extra spec: xyz_in_all + use_tanh + two latent_in layers, plain linears
    dec = Decoder(5, dims=[48] * 5, dropout=None, norm_layers=(), latent_in=[2, 4], weight_norm=False,
                  xyz_in_all=True, use_tanh=True)
    dec.eval()
    inp = (torch.randn(200, 8) * 0.6).requires_grad_(True)
    sdf, scale = dec(inp)
    g_out = torch.randn_like(sdf)
    (sdf * g_out).sum().backward()
    for k, v in state_np(dec).items():
        arrs["x_state_%s" % k] = v
    arrs["x_inputs"] = inp.detach().numpy()
    arrs["x_sdf"] = sdf.detach().numpy()
    arrs["x_gout"] = g_out.numpy()
    arrs["x_grad_inputs"] = inp.grad.numpy()
    # the fitted 8x512 fixture on a strided sample of the D=40 grid
    dec, L = load_fitted()
    g = ref_grid.Grid3D(40, "cpu", torch.float32).points.detach()[::31]
    lat = F.normalize(torch.tensor([0.3, -0.5, 0.8]), p=2, dim=0)
    inp = torch.cat([lat.expand(g.size(0), -1), g], 1).clone().requires_grad_(True)
    sdf, scale = dec(inp)
    sdf.sum().backward()
    arrs["fit_inputs"] = inp.detach().numpy()
    arrs["fit_sdf"] = sdf.detach().numpy()
    arrs["fit_scale"] = scale.detach().numpy()
    arrs["fit_grad_inputs"] = inp.grad.numpy()
    save("g2_decoder.npz", **arrs)


def surface_case(D, latent, dec=None):
    dec = dec or load_fitted()[0]
    grid = ref_grid.Grid3D(D, "cpu", torch.float32)
    lat = torch.tensor(latent, dtype=torch.float32, requires_grad=True)
    lat_ = F.normalize(lat, p=2, dim=0)
    inputs = torch.cat([lat_.expand(grid.points.size(0), -1), grid.points], 1)
    sdf, _ = dec(inputs)
    pts, nocs, nrm = grid.get_surface_points(sdf)
    return dec, grid, lat, sdf, pts, nocs, nrm


def g3():
    arrs = {}
    for tag, D, latent in (("a", 16, [0.3, -0.5, 0.8]), ("b", 21, [-0.6, 0.2, 0.1])):
        dec, grid, lat, sdf, pts, nocs, nrm = surface_case(D, latent)
        s = sdf.detach().numpy()
        arrs[tag + "_D"] = D
        arrs[tag + "_latent"] = np.asarray(latent, np.float32)
        arrs[tag + "_sdf"] = s
        arrs[tag + "_grad_points"] = ref_grid.grads["grid_points"].detach().numpy()  # n_hat after the in-place divide
        arrs[tag + "_points"] = pts.detach().numpy()
        arrs[tag + "_nocs"] = nocs.detach().numpy()
        arrs[tag + "_normals"] = nrm.detach().numpy()
        arrs[tag + "_band_idx"] = np.nonzero(np.abs(s[:, 0]) < 0.03)[0]
        arrs[tag + "_band_margin"] = np.min(np.abs(np.abs(s[:, 0]) - 0.03))
        print("G3", tag, "N =", pts.shape[0], "margin", arrs[tag + "_band_margin"])
    save("g3_surface.npz", **arrs)


def g4():
    arrs = {}
    dec, grid, lat, sdf, pts, nocs, nrm = surface_case(16, [0.3, -0.5, 0.8])
    pts, nrm = pts.detach(), nrm.detach()
    arrs["points"] = pts.numpy()
    arrs["normals"] = nrm.numpy()
    K = K_for(32, 32)
    arrs["K"] = K.numpy()
    for i, (yaw, trans) in enumerate((([0.6], [0.0, 0.0, 3.5]), ([-1.1], [0.2, -0.1, 2.8]), ([2.5], [-0.3, 0.15, 4.2]))):
        pose = build_pose(torch.tensor(yaw), torch.tensor(trans))
        for nocs_flag in (True, False):
            o = ref_proj.project_in_2D(K, pose, pts, nrm, nrm, (32, 32), output_nocs=nocs_flag)
            t = "dcm%d_%s_" % (i, "nocs" if nocs_flag else "col")
            arrs[t + "pose"] = pose.numpy()
            for k, v in o.items():
                arrs[t + k] = v.numpy()
            dot = (o["normals_3d"] * o["points_3d"]).sum(1)
            arrs[t + "filt_margin"] = dot.abs().min().numpy()
    q = F.normalize(torch.tensor([0.9, 0.1, 0.35, -0.2]), dim=0)
    cam = torch.cat([q, torch.tensor([0.1, -0.05, 3.2])])
    with torch.no_grad():
        o = ref_proj.project_in_2D_quat(K, cam, pts, nrm, nrm, (32, 32), output_nocs=True)
    arrs["quat_pose"] = cam.numpy()
    for k, v in o.items():
        arrs["quat_" + k] = v.detach().numpy()
    save("g4_project.npz", **arrs)


def g5():
    arrs = {}
    torch.manual_seed(3)
    H = W = 16
    K = K_for(H, W)
    r = Rasterer(K, (W, H), precision=torch.float32)
    # small synthetic surfel set in front of the camera facing it, dense enough to overlap
    N = 64
    p = torch.stack([torch.rand(N) * 0.5 - 0.25, torch.rand(N) * 0.5 - 0.25, 1.0 + torch.rand(N) * 0.3], 1)
    n = F.normalize(torch.randn(N, 3) * 0.4 + torch.tensor([0, 0, -1.0]), dim=1)
    n[:4] = F.normalize(torch.tensor([[1.0, 0.0, 0.002], [0.0, 1.0, -0.004], [0.7, 0.7, 0.003], [1.0, 0.2, 0.0]]), dim=1)  # grazing
    p2 = torch.zeros(N, 2)
    for bg in (False, True):
        w = ref_prim.inside_surfel(K, r.grid, p2, p, n, diam=0.04, softclamp=False, add_bg=bg)
        arrs["w_bg%d" % int(bg)] = w[:, 0, :].numpy()
        assert torch.equal(w[:, 0], w[:, 1]) and torch.equal(w[:, 0], w[:, 2])
    arrs["K"] = K.numpy()
    arrs["Kinv"] = K.float().inverse().numpy()
    arrs["points"] = p.numpy()
    arrs["normals"] = n.numpy()
    arrs["res"] = np.array([W, H])
    arrs["grid"] = r.grid.numpy()
    save("g5_inside_surfel.npz", **arrs)


def g6():
    arrs = {}
    dec, grid, lat, sdf, pts, nocs, nrm = surface_case(16, [0.3, -0.5, 0.8])
    pts, nrm = pts.detach(), nrm.detach()
    arrs["points"] = pts.numpy()
    arrs["normals"] = nrm.numpy()
    torch.manual_seed(11)
    colors = torch.rand(pts.size(0), 3)
    arrs["colors"] = colors.numpy()
    pose = build_pose(torch.tensor([0.6]), torch.tensor([0.0, 0.0, 3.5]))
    arrs["pose"] = pose.numpy()
    for (H, W) in ((32, 32), (64, 48)):
        K = K_for(H, W)
        r = Rasterer(K, (W, H), precision=torch.float32)
        t0 = "r%dx%d_" % (H, W)
        arrs[t0 + "K"] = K.numpy()
        arrs[t0 + "Kinv"] = K.float().inverse().numpy()
        for nocs_flag in (True, False):
            rend, points = r(pts, nrm, colors, pose, rot="dcm", primitives="disc", bg=None, output_mask=True,
                             output_depth=True, output_normals=True, output_nocs=nocs_flag, output_points=True)
            t = t0 + ("nocs_" if nocs_flag else "col_")
            for k, v in rend.items():
                arrs[t + k] = v.numpy()
            for k, v in points.items():
                arrs[t + "pts_" + k] = v.numpy()
        bg = torch.rand(3, H, W)
        rend = r(pts, nrm, colors, pose, rot="dcm", primitives="disc", bg=bg, output_mask=True,
                 output_depth=False, output_normals=False, output_nocs=True, output_points=False)
        arrs[t0 + "bg"] = bg.numpy()
        for k, v in rend.items():
            arrs[t0 + "bg_" + k] = v.numpy()
    save("g6_rasterer.npz", **arrs)


def g7():
    """End-to-end autograd gradients through the optimizer's graph (optimizer.py:79-123), losses replaced by a
    fixed random linear functional of every differentiable output."""
    arrs = {}
    for tag, D, H, W, latent, yaw0, trans0 in (("a", 16, 32, 32, [0.3, -0.5, 0.8], 0.6, [0.0, 0.0, 3.5]),
                                               ("b", 21, 48, 40, [-0.6, 0.2, 0.1], -0.9, [0.15, -0.1, 3.0])):
        dec = load_fitted()[0]
        grid = ref_grid.Grid3D(D, "cpu", torch.float32)
        lat = torch.tensor(latent, dtype=torch.float32, requires_grad=True)
        yaw = torch.tensor([yaw0], requires_grad=True)
        trans = torch.tensor(trans0, requires_grad=True)
        K = K_for(H, W)
        renderer = Rasterer(K, (W, H), precision=torch.float32)
        lat_ = F.normalize(lat, p=2, dim=0)
        inputs = torch.cat([lat_.expand(grid.points.size(0), -1), grid.points], 1)
        sdf, _ = dec(inputs)
        pcd, _, normals = grid.get_surface_points(sdf)
        lat.grad = None
        dec.zero_grad()
        grid.points.grad = None
        pcd.retain_grad()
        pose = build_pose(yaw, trans)
        pose.retain_grad()
        rendering, points = renderer(pcd, normals, normals, pose, primitives="disc", rot="dcm", bg=None,
                                     output_depth=True, output_normals=True, output_nocs=True, output_points=True,
                                     output_mask=True)
        gen = torch.Generator().manual_seed(5)
        Wt = {k: torch.randn(v.shape, generator=gen) for k, v in rendering.items()}
        Wp = {k: torch.randn(points[k].shape, generator=gen) for k in ("xyzf", "rgbf", "xyz", "rgb")}
        loss = sum((rendering[k] * Wt[k]).sum() for k in rendering) + sum((points[k] * Wp[k]).sum() for k in Wp)
        loss.backward()
        arrs[tag + "_cfg"] = np.array([D, H, W])
        arrs[tag + "_latent"] = np.asarray(latent, np.float32)
        arrs[tag + "_yaw"] = np.asarray([yaw0], np.float32)
        arrs[tag + "_trans"] = np.asarray(trans0, np.float32)
        arrs[tag + "_K"] = K.numpy()
        arrs[tag + "_Kinv"] = K.float().inverse().numpy()
        arrs[tag + "_sdf"] = sdf.detach().numpy()
        arrs[tag + "_pcd"] = pcd.detach().numpy()
        arrs[tag + "_normals"] = normals.detach().numpy()
        arrs[tag + "_pose"] = pose.detach().numpy()
        for k, v in rendering.items():
            arrs[tag + "_out_" + k] = v.detach().numpy()
            arrs[tag + "_W_" + k] = Wt[k].numpy()
        for k in Wp:
            arrs[tag + "_pts_" + k] = points[k].detach().numpy()
            arrs[tag + "_Wp_" + k] = Wp[k].numpy()
        arrs[tag + "_loss"] = loss.detach().numpy()
        arrs[tag + "_g_yaw"] = yaw.grad.numpy()
        arrs[tag + "_g_trans"] = trans.grad.numpy()
        arrs[tag + "_g_latent"] = lat.grad.numpy()
        arrs[tag + "_g_pcd"] = pcd.grad.numpy()
        arrs[tag + "_g_pose"] = pose.grad.numpy()
        arrs[tag + "_g_gridpoints_absmax"] = grid.points.grad.abs().max().numpy()
        print("G7", tag, "N", pcd.shape[0], "loss", float(loss), "g_yaw", yaw.grad.numpy(), "g_lat", lat.grad.numpy())
    save("g7_grads.npz", **arrs)


def synth_targets(dec, D, H, W, latent_gt, yaw_gt, trans_gt, scale_gt):
    """Render the GT pose with the reference renderer to obtain a target NOCS image and a lidar-like cloud
    (the a-harness of SURVEY.md §8: what refine_css_demo.py:107-131 would supply from the CSS net + lidar)."""
    grid = ref_grid.Grid3D(D, "cpu", torch.float32)
    lat_ = F.normalize(torch.tensor(latent_gt), p=2, dim=0)
    inputs = torch.cat([lat_.expand(grid.points.size(0), -1), grid.points], 1)
    sdf, _ = dec(inputs)
    pcd, _, normals = grid.get_surface_points(sdf)
    pose = build_pose(torch.tensor([yaw_gt]), torch.tensor(trans_gt))
    K = K_for(H, W)
    renderer = Rasterer(K, (W, H), precision=torch.float32)
    rendering, points = renderer(pcd.detach(), normals.detach(), normals.detach(), pose, primitives="disc", rot="dcm",
                                 output_nocs=True, output_points=True, output_mask=True)
    nocs = rendering["color"].detach()
    lidar = (points["xyzf"].detach() * scale_gt)[::3].numpy().copy()
    return K, nocs, lidar


def g8(name="g8_optimizer.npz", D=20, H=32, W=32, iters=10):
    from pipelines.optimizer import Optimizer
    dec = load_fitted()[0]
    K, nocs, lidar = synth_targets(dec, D, H, W, [0.3, -0.5, 0.8], 0.6, [0.0, 0.0, 3.5], 2.0)
    params = {"yaw": [0.7], "trans": [0.03, 0.02, 3.45], "scale": [2.0], "latent": [0.5, -0.3, 0.6]}
    opt = Optimizer({k: list(v) for k, v in params.items()}, "cpu", {"2d": 0.3, "3d": 0.5})
    grid = ref_grid.Grid3D(D, "cpu", torch.float32)
    traj = []
    buf = io.StringIO()
    for it in range(iters):
        with contextlib.redirect_stdout(buf):
            opt.optimize(1, nocs, lidar, dec, grid, K, [H, W], viz_type=None)
        traj.append(np.concatenate([opt.params[k].detach().numpy().ravel() for k in ("yaw", "trans", "scale", "latent")]))
    losses = [l for l in buf.getvalue().splitlines() if l.startswith("ITER")]
    l2d, l3d = [], []
    for l in losses:
        parts = l.split("2D - ")[1].split(", 3D - ")
        l2d.append(float(parts[0]))
        l3d.append(float(parts[1].split(", Total")[0]))
    print("G8 losses2d", l2d[:3], "...", l2d[-1], "traj yaw", [t[0] for t in traj])
    save(name, D=D, H=H, W=W, K=K.numpy(), nocs_target=nocs.numpy(), lidar=lidar,
         init=np.concatenate([np.asarray(params[k], np.float32) for k in ("yaw", "trans", "scale", "latent")]),
         traj=np.asarray(traj), loss2d_weighted=np.asarray(l2d), loss3d_weighted=np.asarray(l3d))


def g8b():
    """The same 10-iteration Optimizer trajectory at BASELINE configs[0]'s size: one 128x128 crop, D = 40 (the reference's dense 2-D loss
    needs O(rendered pixels x H W) temporaries: ~2 GB here, out of reach at 256x256)."""
    g8("g8b_optimizer_128.npz", D=40, H=128, W=128)


def g8c():
    """The reference's full refinement length (60 iterations, config_refine.ini:15) at the G8 size: where the reference converges to."""
    g8("g8c_optimizer_60it.npz", iters=60)


def g8h():
    """The reference Optimizer in its shipped precision (config_refine.ini:19 float16: decoder, grid, K, target NOCS all half, as
    refine_css.py:144-153 builds them) on the G8 problem: 10 iterations.  The float32 run of the same problem is golden G8; the gap between
    the two is the yardstick of the tolerance stated in tests/test_gpu_configs.py."""
    from pipelines.optimizer import Optimizer
    D, H, W = 20, 32, 32
    K, nocs, lidar = synth_targets(load_fitted()[0], D, H, W, [0.3, -0.5, 0.8], 0.6, [0.0, 0.0, 3.5], 2.0)
    dec16 = load_fitted(torch.float16)[0]
    params = {"yaw": [0.7], "trans": [0.03, 0.02, 3.45], "scale": [2.0], "latent": [0.5, -0.3, 0.6]}
    opt = Optimizer({k: list(v) for k, v in params.items()}, "cpu", {"2d": 0.3, "3d": 0.5})
    grid = ref_grid.Grid3D(D, "cpu", torch.float16)
    traj, buf = [], io.StringIO()
    for it in range(10):
        with contextlib.redirect_stdout(buf):
            opt.optimize(1, nocs.half(), lidar, dec16, grid, K.half(), [H, W], viz_type=None)
        traj.append(np.concatenate([opt.params[k].detach().float().numpy().ravel() for k in ("yaw", "trans", "scale", "latent")]))
    l2d, l3d = [], []
    for l in (l for l in buf.getvalue().splitlines() if l.startswith("ITER")):
        parts = l.split("2D - ")[1].split(", 3D - ")
        l2d.append(float(parts[0])); l3d.append(float(parts[1].split(", Total")[0]))
    print("G8h traj yaw", [t[0] for t in traj])
    save("g8h_optimizer_fp16.npz", D=D, H=H, W=W, K=K.numpy(), nocs_target=nocs.numpy(), lidar=lidar,
         init=np.concatenate([np.asarray(params[k], np.float32) for k in ("yaw", "trans", "scale", "latent")]),
         traj=np.asarray(traj), loss2d_weighted=np.asarray(l2d), loss3d_weighted=np.asarray(l3d))


def g8s():
    """The Optimizer's skip rules and degenerate losses (optimizer.py:127-129 no surfels / no lidar -> 'Skip frame'; :149-151 NaN or zero
    loss -> 'Skip frame'; compute_loss_3d without close pairs -> 0): four iterations each of
      far    lidar 5 m away: no pair within the threshold, the 3-D loss is 0 and the step uses the 2-D loss alone
      nolidar  empty lidar array: every iteration skipped, parameters untouched
      blank  target NOCS image all zero and far lidar: what the reference does with an empty 2-D target"""
    from pipelines.optimizer import Optimizer
    D, H, W = 20, 32, 32
    dec = load_fitted()[0]
    K, nocs, lidar = synth_targets(dec, D, H, W, [0.3, -0.5, 0.8], 0.6, [0.0, 0.0, 3.5], 2.0)
    init = {"yaw": [0.7], "trans": [0.03, 0.02, 3.45], "scale": [2.0], "latent": [0.5, -0.3, 0.6]}
    cases = {"far": (nocs, lidar + np.array([[5.0, 0.0, 0.0]], np.float32)), "nolidar": (nocs, np.zeros((0, 3), np.float32)),
             "blank": (torch.zeros_like(nocs), lidar + np.array([[5.0, 0.0, 0.0]], np.float32))}
    arrs = dict(D=D, H=H, W=W, K=K.numpy(), nocs_target=nocs.numpy(), lidar=lidar,
                init=np.concatenate([np.asarray(init[k], np.float32) for k in ("yaw", "trans", "scale", "latent")]))
    for tag, (tgt, ld) in cases.items():
        opt = Optimizer({k: list(v) for k, v in init.items()}, "cpu", {"2d": 0.3, "3d": 0.5})
        grid = ref_grid.Grid3D(D, "cpu", torch.float32)
        traj, lines = [], []
        for it in range(4):
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                opt.optimize(1, tgt, ld, dec, grid, K, [H, W], viz_type=None)
            lines.append(buf.getvalue().strip().splitlines()[-1] if buf.getvalue().strip() else "")
            traj.append(np.concatenate([opt.params[k].detach().numpy().ravel() for k in ("yaw", "trans", "scale", "latent")]))
        print("G8s", tag, lines, [float(t[0]) for t in traj])
        arrs[tag + "_traj"] = np.asarray(traj)
        arrs[tag + "_skipped"] = np.array([1 if l.startswith("Skip") else 0 for l in lines], np.int32)
        arrs[tag + "_lidar"] = ld
        arrs[tag + "_target"] = tgt.numpy()
    save("g8s_optimizer_skips.npz", **arrs)


def g9():
    """Secondary rows a6' / bg: Rasterer.forward with primitives circle / circle_opt and with a background image, plus autograd
    gradients w.r.t. the surfel positions and the pose."""
    arrs = {}
    dec, grid, lat, sdf, pts, nocs, nrm = surface_case(16, [0.3, -0.5, 0.8])
    pts0, nrm0 = pts.detach(), nrm.detach()
    arrs["points"] = pts0.numpy()
    arrs["normals"] = nrm0.numpy()
    H, W = 32, 32
    K = K_for(H, W)
    arrs["K"] = K.numpy()
    arrs["Kinv"] = K.float().inverse().numpy()
    r = Rasterer(K, (W, H), precision=torch.float32)
    gen = torch.Generator().manual_seed(17)
    bgimg = torch.rand(3, H, W, generator=gen)
    arrs["bg"] = bgimg.numpy()
    for prim in ("circle", "circle_opt", "disc"):
        for use_bg in (False, True):
            if prim == "disc" and not use_bg:
                continue
            p = pts0.clone().requires_grad_(True)
            yaw = torch.tensor([0.6], requires_grad=True)
            trans = torch.tensor([0.05, -0.03, 3.4], requires_grad=True)
            pose = build_pose(yaw, trans)
            rend = r(p, nrm0, nrm0, pose, rot="dcm", primitives=prim, bg=bgimg if use_bg else None, output_mask=True,
                     output_depth=not use_bg, output_normals=not use_bg, output_nocs=True, output_points=False)
            t = "%s_bg%d_" % (prim, int(use_bg))
            Ws = {k: torch.randn(v.shape, generator=gen) for k, v in rend.items()}
            loss = sum((rend[k] * Ws[k]).sum() for k in rend)
            loss.backward()
            for k, v in rend.items():
                arrs[t + "out_" + k] = v.detach().numpy()
                arrs[t + "W_" + k] = Ws[k].numpy()
            arrs[t + "pose"] = pose.detach().numpy()
            arrs[t + "g_points"] = p.grad.numpy()
            arrs[t + "g_yaw"] = yaw.grad.numpy()
            arrs[t + "g_trans"] = trans.grad.numpy()
            print("G9", t, "loss", float(loss), "g_yaw", yaw.grad.numpy(), "|g_points|max", float(p.grad.abs().max()))
    save("g9_secondary.npz", **arrs)


def pattern_weights(shape, salt):
    """Deterministic pseudo-random weights in [-1, 1] (integer hash of the flat index: exact in float32 on every platform), so that a
    gradient functional over large outputs needs no stored weight arrays.  tests/_util.py holds the same function."""
    n = int(np.prod(shape))
    i = np.arange(n, dtype=np.uint64)
    h = (i * np.uint64(2654435761) + np.uint64(salt) * np.uint64(40503)) & np.uint64(0xFFFFFFFF)
    h = (h >> np.uint64(7)) % np.uint64(2001)
    return ((h.astype(np.int64) - 1000).astype(np.float32) / np.float32(1000.0)).reshape(shape)


def near_threshold_pixels(K, H, W, pose, pcd, normals):
    """Pixels where some (pixel, surfel) pair sits within 1e-5 of a selection threshold of inside_surfel (disc edge, |n.ray| = 0.01):
    computed with THIS project's oracle on the reference's surfels, stored so that a GPU test can attribute a rounding flip."""
    sys.path.insert(0, os.path.join(HERE, ".."))
    from oracle import sdf_oracle as O
    Kn = K.numpy().astype(np.float32)
    Kinv = np.linalg.inv(Kn).astype(np.float32)
    proj = O.project_in_2D(Kn, pose, pcd, normals, normals, (W, H), output_nocs=True)
    v3, nc = proj["points_3d"].astype(np.float32), proj["normals_3d"].astype(np.float32)
    near = np.zeros(H * W, bool)
    grid2d = O.pixel_grid((W, H)).reshape(H, W, 2)
    for r0 in range(0, H, 32):
        sub = grid2d[r0:r0 + 32].reshape(-1, 2)
        _, aux = O.inside_surfel(Kinv, sub, v3, nc, diam=0.04, want_aux=True)
        near[r0 * W:r0 * W + sub.shape[0]] = (aux["margin_disc"] < 1e-5) | (aux["margin_b"] < 1e-5)
    return near


def g10(name="g10_config1_256.npz", latent=(0.5, -0.3, 0.6), yaw0=0.7, trans0=(0.03, 0.02, 3.45)):
    """BASELINE configs[1] at its stated size: ONE 256x256 crop, D = 40, float32, the optimizer's graph (optimizer.py:79-123) with every
    output enabled and a deterministic linear functional as the loss; images, surfels and autograd gradients of the reference."""
    D, H, W = 40, 256, 256
    latent, trans0 = list(latent), list(trans0)
    dec = load_fitted()[0]
    grid = ref_grid.Grid3D(D, "cpu", torch.float32)
    lat = torch.tensor(latent, dtype=torch.float32, requires_grad=True)
    yaw = torch.tensor([yaw0], requires_grad=True)
    trans = torch.tensor(trans0, requires_grad=True)
    K = K_for(H, W)
    renderer = Rasterer(K, (W, H), precision=torch.float32)
    lat_ = F.normalize(lat, p=2, dim=0)
    inputs = torch.cat([lat_.expand(grid.points.size(0), -1), grid.points], 1)
    sdf, _ = dec(inputs)
    pcd, _, normals = grid.get_surface_points(sdf)
    lat.grad = None
    dec.zero_grad()
    grid.points.grad = None
    pose = build_pose(yaw, trans)
    rendering, points = renderer(pcd, normals, normals, pose, primitives="disc", rot="dcm", bg=None, output_depth=True,
                                 output_normals=True, output_nocs=True, output_points=True, output_mask=True)
    salts = {"color": 1, "mask": 2, "depth": 3, "normals": 4, "xyzf": 5}
    loss = sum((rendering[k] * torch.from_numpy(pattern_weights(tuple(rendering[k].shape), salts[k]))).sum() for k in rendering)
    loss = loss + (points["xyzf"] * torch.from_numpy(pattern_weights(tuple(points["xyzf"].shape), salts["xyzf"]))).sum()
    loss.backward()
    s = sdf.detach().numpy()
    arrs = dict(cfg=np.array([D, H, W]), latent=np.asarray(latent, np.float32), yaw=np.asarray([yaw0], np.float32),
                trans=np.asarray(trans0, np.float32), K=K.numpy(), pose=pose.detach().numpy(),
                sdf_stride7=s[::7, 0], band_idx=np.nonzero(np.abs(s[:, 0]) < 0.03)[0].astype(np.int32),
                band_margin=np.min(np.abs(np.abs(s[:, 0]) - 0.03)), pcd=pcd.detach().numpy(), normals=normals.detach().numpy(),
                xyzf=points["xyzf"].detach().numpy(), loss=loss.detach().numpy(), g_yaw=yaw.grad.numpy(), g_trans=trans.grad.numpy(),
                g_latent=lat.grad.numpy())
    dot = ((normals.detach() @ pose.detach()[:3, :3].t()) * points["xyz"].detach()).sum(1)
    arrs["filt_margin"] = dot.abs().min().numpy()
    for k, v in rendering.items():
        arrs["out_" + k] = v.detach().numpy()
    arrs["near_threshold"] = np.packbits(near_threshold_pixels(K, H, W, pose.detach().numpy(), pcd.detach().numpy(), normals.detach().numpy()))
    print("G10 N", pcd.shape[0], "Nf", points["xyzf"].shape[0], "loss", float(loss), "g_yaw", yaw.grad.numpy(), "g_trans", trans.grad.numpy(),
          "g_lat", lat.grad.numpy(), "band margin", arrs["band_margin"], "filt margin", arrs["filt_margin"],
          "near px", int(np.unpackbits(arrs["near_threshold"]).sum()))
    save(name, **arrs)


def g10b():
    """a second crop at the configs[1] size: another shape, a side view from closer up (larger surfels, more of them per pixel)"""
    g10("g10b_config1_256.npz", latent=(-0.6, 0.2, 0.1), yaw0=-1.1, trans0=(0.2, -0.1, 2.9))


def _ref_render_precision(prec, D, H, W, latent, yaw0, trans0):
    """the reference's pipeline at one `precision` (config_refine.ini:19 -> refine_css.py:144-153: decoder, grid, K, pose, Rasterer all in it)"""
    dec, _ = load_fitted(prec)
    grid = ref_grid.Grid3D(D, "cpu", prec)
    lat_ = F.normalize(torch.tensor(latent).to(prec), p=2, dim=0)                                    # optimizer.py:96
    inputs = torch.cat([lat_.expand(grid.points.size(0), -1), grid.points], 1).to(lat_.dtype)       # :99-100
    sdf, _ = dec(inputs)
    pcd, _, normals = grid.get_surface_points(sdf)
    pose = torch.eye(4).to(prec)                                                                      # :86-90
    pose[:3, :3] = rtools.rot_from_yaw(torch.tensor([yaw0])).to(prec)
    pose[1] *= -1
    pose[:3, 3] = torch.tensor(trans0).to(prec)
    K = K_for(H, W).to(prec)
    r = Rasterer(K, (W, H), precision=prec)
    with torch.no_grad():
        rend, pts = r(pcd.detach(), normals.detach(), normals.detach(), pose, primitives="disc", rot="dcm", bg=None, output_depth=True,
                      output_normals=True, output_nocs=True, output_points=True, output_mask=True)
    return sdf.detach(), pcd.detach(), normals.detach(), rend, pts


def g11():
    """BASELINE configs[4]: the reference's OWN float16 run (setup_dsdf(precision=float16), Grid3D(..., float16), half K / pose / Rasterer,
    configs/config_refine.ini:19) at 512x512, D = 40, beside its float32 run of the same inputs.  Stored: the float16 decoder output on the
    whole grid and its band, the float16 images (as float16, exact), the float32 images, and the per-image counts of pixels where the
    reference's two precisions disagree by more than 1e-2 -- the yardstick for the tolerance stated in tests/test_gpu_configs.py."""
    D, H, W = 40, 512, 512
    latent, yaw0, trans0 = [0.3, -0.5, 0.8], 0.6, [0.0, 0.0, 3.5]
    arrs = dict(cfg=np.array([D, H, W]), latent=np.asarray(latent, np.float32), yaw=np.asarray([yaw0], np.float32),
                trans=np.asarray(trans0, np.float32), K=K_for(H, W).numpy())
    res = {}
    for tag, prec in (("f32", torch.float32), ("f16", torch.float16)):
        sdf, pcd, normals, rend, pts = _ref_render_precision(prec, D, H, W, latent, yaw0, trans0)
        res[tag] = (sdf.float().numpy(), {k: v.float().numpy() for k, v in rend.items()})
        s = sdf.float().numpy()[:, 0]
        store = np.float16 if prec == torch.float16 else np.float32
        arrs[tag + "_sdf"] = sdf.numpy()[:, 0].astype(store)
        arrs[tag + "_band_idx"] = np.nonzero(np.abs(s) < 0.03)[0].astype(np.int32)
        arrs[tag + "_n_front"] = pts["xyzf"].shape[0]
        for k, v in rend.items():
            arrs[tag + "_out_" + k] = v.numpy().astype(store)
    for k in res["f32"][1]:
        d = np.abs(res["f32"][1][k] - res["f16"][1][k])
        arrs["ref_pixels_beyond_1e-2_" + k] = int((d.reshape(d.shape[0], -1).max(0) > 1e-2).sum())
        print("G11", k, "reference f16 vs f32: pixels beyond 1e-2:", arrs["ref_pixels_beyond_1e-2_" + k], "of", H * W)
    arrs["ref_sdf_max_abs_diff"] = float(np.abs(res["f32"][0] - res["f16"][0]).max())
    arrs["ref_sdf_mean_abs_diff"] = float(np.abs(res["f32"][0] - res["f16"][0]).mean())
    b32, b16 = set(arrs["f32_band_idx"].tolist()), set(arrs["f16_band_idx"].tolist())
    arrs["ref_band_symmetric_difference"] = len(b32 ^ b16)
    print("G11 sdf max/mean diff", arrs["ref_sdf_max_abs_diff"], arrs["ref_sdf_mean_abs_diff"], "band", len(b32), len(b16), "sym diff", len(b32 ^ b16))
    save("g11_config4_fp16_512.npz", **arrs)


def g12():
    """Standalone goldens of the two losses (pipelines/optimizer.py:166-237): values and autograd gradients w.r.t. the rendered NOCS image,
    the estimated points and the scale (through pcd_frustum = lidar / scale, optimizer.py:84)."""
    from pipelines.optimizer import Optimizer
    arrs = {}
    dec = load_fitted()[0]
    for tag, D, H, W in (("a", 20, 32, 32), ("b", 27, 40, 56)):
        K, nocs_t, lidar = synth_targets(dec, D, H, W, [0.3, -0.5, 0.8], 0.6, [0.0, 0.0, 3.5], 2.0)
        # the rendering of a perturbed pose / shape = what the loss sees in the first iteration
        grid = ref_grid.Grid3D(D, "cpu", torch.float32)
        lat_ = F.normalize(torch.tensor([0.5, -0.3, 0.6]), p=2, dim=0)
        inputs = torch.cat([lat_.expand(grid.points.size(0), -1), grid.points], 1)
        sdf, _ = dec(inputs)
        pcd, _, normals = grid.get_surface_points(sdf)
        pose = build_pose(torch.tensor([0.7]), torch.tensor([0.03, 0.02, 3.45]))
        renderer = Rasterer(K, (W, H), precision=torch.float32)
        rendering, points = renderer(pcd.detach(), normals.detach(), normals.detach(), pose, primitives="disc", rot="dcm", output_nocs=True,
                                     output_points=True, output_mask=True)
        opt = Optimizer({"yaw": [0.7], "trans": [0.03, 0.02, 3.45], "scale": [2.0], "latent": [0.5, -0.3, 0.6]}, "cpu", {"2d": 0.3, "3d": 0.5})
        opt.device, opt.precision = torch.device("cpu"), torch.float32
        for thr_tag, thr in (("", 1.0), ("_t03", 0.3)):
            col = rendering["color"].detach().clone().requires_grad_(True)
            l2 = opt.compute_loss_2d(col, nocs_t, diam=5, threshold_nocs=thr)
            l2.backward()
            arrs[tag + "_l2d" + thr_tag] = l2.detach().numpy()
            arrs[tag + "_g_color" + thr_tag] = col.grad.numpy()
        est = points["xyzf"].detach().clone().requires_grad_(True)
        scale = opt.params["scale"]
        frustum = torch.Tensor(lidar) / scale                                   # optimizer.py:84
        l3, dists, idxs = opt.compute_loss_3d(est, frustum)
        l3.backward()
        arrs[tag + "_cfg"] = np.array([D, H, W])
        arrs[tag + "_color"] = rendering["color"].detach().numpy()
        arrs[tag + "_target"] = nocs_t.numpy()
        arrs[tag + "_xyzf"] = est.detach().numpy()
        arrs[tag + "_lidar"] = lidar
        arrs[tag + "_scale"] = np.asarray([2.0], np.float32)
        arrs[tag + "_l3d"] = l3.detach().numpy()
        arrs[tag + "_g_xyzf"] = est.grad.numpy()
        arrs[tag + "_g_scale"] = scale.grad.numpy()
        arrs[tag + "_nn_idx"] = np.asarray(idxs, np.int32)
        arrs[tag + "_n_pairs"] = int((np.asarray(dists) < 0.2 / 2.0).sum())
        print("G12", tag, "l2d", float(arrs[tag + "_l2d"]), "l2d(thr .3)", float(arrs[tag + "_l2d_t03"]), "l3d", float(l3), "pairs", arrs[tag + "_n_pairs"], "of", est.shape[0],
              "g_scale", scale.grad.numpy())
    save("g12_losses.npz", **arrs)


def g13():
    """The three standalone primitives (primitives.py:4-242) as Rasterer.forward calls them (rasterer.py:92-104): dense weight matrices
    and autograd gradients of a fixed random functional w.r.t. the camera-frame vertices and normals; plus project_in_2D(_quat) gradients."""
    arrs = {}
    torch.manual_seed(13)
    H, W = 16, 24
    K = K_for(H, W)
    K[0, 2], K[1, 2] = W / 2.0, H / 2.0
    r = Rasterer(K, (W, H), precision=torch.float32)
    N = 48
    p = torch.stack([torch.rand(N) * 0.7 - 0.35, torch.rand(N) * 0.5 - 0.25, 1.0 + torch.rand(N) * 0.4], 1)
    n = F.normalize(torch.randn(N, 3) * 0.4 + torch.tensor([0, 0, -1.0]), dim=1)
    eps = torch.finfo(torch.float32).eps
    h2 = (K @ p.t()).t()
    uv = h2[:, :2] / (h2[:, 2:] + eps)
    uv = torch.cat([torch.clamp(uv[:, 0:1], -1, W), torch.clamp(uv[:, 1:2], -1, H)], -1)
    arrs.update(K=K.numpy(), points=p.numpy(), normals=n.numpy(), uv=uv.numpy(), res=np.array([W, H]))
    gen = torch.Generator().manual_seed(14)
    for name in ("disc", "circle", "circle_opt"):
        for bg in (False, True):
            pp = p.clone().requires_grad_(True)
            nn_ = n.clone().requires_grad_(True)
            if name == "disc":
                w = ref_prim.inside_surfel(K, r.grid, uv, pp, nn_, diam=0.04, softclamp=False, add_bg=bg)
            elif name == "circle":
                w = ref_prim.inside_circle(K, r.grid, uv, pp, nn_, diam=0.02, add_bg=bg)
            else:
                w = ref_prim.inside_circle_opt(K, r.grid_prim, uv, pp, nn_, diam=0.025, add_bg=bg)
            R = torch.randn(w.shape[0], w.shape[2], generator=gen)
            (w[:, 0, :] * R).sum().backward()
            t = "%s_bg%d_" % (name, int(bg))
            arrs[t + "w"] = w[:, 0, :].detach().numpy()
            arrs[t + "R"] = R.numpy()
            arrs[t + "g_points"] = pp.grad.numpy()
            arrs[t + "g_normals"] = nn_.grad.numpy() if nn_.grad is not None else np.zeros((N, 3), np.float32)
            print("G13", t, "covered pairs", int((w[:, 0, :] > 0).sum()), "|g_p|max", float(pp.grad.abs().max()))
    # projection gradients (object-frame inputs)
    dec, grid, lat, sdf, pts, nocs, nrm = surface_case(16, [0.3, -0.5, 0.8])
    pts0, nrm0 = pts.detach()[::3].clone(), nrm.detach()[::3].clone()
    arrs["proj_points"], arrs["proj_normals"] = pts0.numpy(), nrm0.numpy()
    K2 = K_for(32, 32)
    arrs["proj_K"] = K2.numpy()
    for tag in ("dcm", "quat"):
        pp = pts0.clone().requires_grad_(True)
        nn_ = nrm0.clone().requires_grad_(True)
        if tag == "dcm":
            yaw = torch.tensor([0.6], requires_grad=True)
            trans = torch.tensor([0.05, -0.03, 3.4], requires_grad=True)
            cam = build_pose(yaw, trans)
            cam.retain_grad()
            o = ref_proj.project_in_2D(K2, cam, pp, nn_, nn_, (32, 32), output_nocs=True)
        else:
            cam = torch.cat([F.normalize(torch.tensor([0.9, 0.1, 0.35, -0.2]), dim=0), torch.tensor([0.1, -0.05, 3.2])]).requires_grad_(True)
            o = ref_proj.project_in_2D_quat(K2, cam, pp, nn_, nn_, (32, 32), output_nocs=True)
        loss = 0
        for k in sorted(o):
            Wk = torch.randn(o[k].shape, generator=gen)
            arrs["proj_%s_W_%s" % (tag, k)] = Wk.numpy()
            arrs["proj_%s_out_%s" % (tag, k)] = o[k].detach().numpy()
            loss = loss + (o[k] * Wk).sum()
        loss.backward()
        arrs["proj_%s_cam" % tag] = cam.detach().numpy()
        arrs["proj_%s_g_cam" % tag] = cam.grad.numpy()
        arrs["proj_%s_g_points" % tag] = pp.grad.numpy()
        arrs["proj_%s_g_normals" % tag] = nn_.grad.numpy()
        print("G13 proj", tag, "keys", sorted(o), "loss", float(loss))
    save("g13_primitives.npz", **arrs)


def g13s():
    """The standalone primitives in the clamp configurations Rasterer.forward does NOT use (VERDICT r04 missing 4): inside_surfel with its
    own defaults (softclamp=True, softclamp_constant=5, diam=0.03, add_bg=True: primitives.py:165-176,213-214) -- the sigmoid is positive
    until exp overflows, so practically every surfel "covers" every pixel --, inside_circle(softclamp=False) (:43-46, a hard circle),
    inside_circle with another softclamp_constant, inside_circle_opt(softclamp=False) (:118-119).  Same inputs as G13; weights and the
    autograd gradients of a fixed random functional."""
    z13 = np.load(os.path.join(OUT, "g13_primitives.npz"))
    K = torch.from_numpy(z13["K"])
    W, H = [int(v) for v in z13["res"]]
    r = Rasterer(K, (W, H), precision=torch.float32)
    p, n, uv = torch.from_numpy(z13["points"]), torch.from_numpy(z13["normals"]), torch.from_numpy(z13["uv"])
    N = p.shape[0]
    gen = torch.Generator().manual_seed(15)
    arrs = {}
    cases = {
        "disc_default_bg1": lambda pp, nn_: ref_prim.inside_surfel(K, r.grid, uv, pp, nn_),
        "disc_soft_bg0": lambda pp, nn_: ref_prim.inside_surfel(K, r.grid, uv, pp, nn_, diam=0.04, softclamp=True, add_bg=False),
        "disc_soft_c40_bg1": lambda pp, nn_: ref_prim.inside_surfel(K, r.grid, uv, pp, nn_, diam=0.04, softclamp=True, softclamp_constant=40, add_bg=True),
        "circle_hard_bg0": lambda pp, nn_: ref_prim.inside_circle(K, r.grid, uv, pp, nn_, diam=0.02, softclamp=False, add_bg=False),
        "circle_hard_bg1": lambda pp, nn_: ref_prim.inside_circle(K, r.grid, uv, pp, nn_, diam=0.02, softclamp=False, add_bg=True),
        "circle_c30_default_diam_bg0": lambda pp, nn_: ref_prim.inside_circle(K, r.grid, uv, pp, nn_, softclamp_constant=30),
        "circle_opt_hard_bg0": lambda pp, nn_: ref_prim.inside_circle_opt(K, r.grid_prim, uv, pp, nn_, diam=0.025, softclamp=False, add_bg=False),
        "circle_opt_hard_bg1": lambda pp, nn_: ref_prim.inside_circle_opt(K, r.grid_prim, uv, pp, nn_, diam=0.025, softclamp=False, add_bg=True),
    }
    for t, fn in cases.items():
        pp = p.clone().requires_grad_(True)
        nn_ = n.clone().requires_grad_(True)
        w = fn(pp, nn_)
        R = torch.randn(w.shape[0], w.shape[2], generator=gen)
        (w[:, 0, :] * R).sum().backward()
        arrs[t + "_w"] = w[:, 0, :].detach().numpy()
        arrs[t + "_R"] = R.numpy()
        arrs[t + "_g_points"] = pp.grad.numpy()
        arrs[t + "_g_normals"] = nn_.grad.numpy() if nn_.grad is not None else np.zeros((N, 3), np.float32)
        wv = w[:, 0, :].detach()
        print("G13s", t, "rows", w.shape[0], "positive entries", int((wv > 0).sum()), "of", wv.numel(), "finite", bool(torch.isfinite(wv).all()),
              "|g_p|max", float(pp.grad.abs().max()), "|g_n|max", float(arrs[t + "_g_normals"].__abs__().max()))
    save("g13s_primitive_clamps.npz", **arrs)



# ---------------------------------------------------------------------------------------------------------
# G14: the camera regime of the reference PIPELINE.  Every golden above uses K_for(): fx = fy, principal point at the crop centre, object on
# the optical axis.  The pipeline never renders like that: refine_css_demo.py:85-95 cuts the 2-D box out of a 1242x375 KITTI frame and
# utils/refinement.py:586-609 (adjust_intrinsics_crop) shifts the principal point by the box corner and rescales the focal lengths to the
# rendering area, so a crop's principal point lies far outside the crop and the object metres off the optical axis.

KITTI_K = [[721.5377, 0.0, 609.5593], [0.0, 721.5377, 172.854], [0.0, 0.0, 1.0]]          # P2 of the KITTI-3D calibration files
KITTI_K_ANISO = [[721.5377, 0.0, 609.5593], [0.0, 707.0493, 180.384], [0.0, 0.0, 1.0]]    # fx != fy (other datasets; exercises both focal lengths)


def kitti_crop_intrinsics(K_full, yaw, trans, max_crop_area, half_extents=(0.56, 0.46, 1.0)):
    """A KITTI-like 2-D box of an object at renderer-space pose (yaw, trans) -- the image of the object's bounding cuboid under the full-frame
    intrinsics, integer corners as the label files have them (refine_css_demo.py:80-82) -- and the crop intrinsics the REFERENCE's own
    adjust_intrinsics_crop derives from it.  Returns crop_size [H, W] (python ints), K_crop (3,3 float32 tensor), bbox."""
    pose = build_pose(torch.tensor([yaw]), torch.tensor(trans)).numpy()
    hx, hy, hz = half_extents
    corners = np.array([[sx * hx, sy * hy, sz * hz, 1.0] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)], np.float64)
    cam = (pose[:3].astype(np.float64) @ corners.T).T
    uv = (np.asarray(K_full, np.float64) @ cam.T).T
    uv = uv[:, :2] / uv[:, 2:]
    l, t = np.floor(uv.min(0)).astype(int)
    r, b = np.ceil(uv.max(0)).astype(int)
    bbox = [int(l), int(t), int(r), int(b)]
    crop_size = torch.Tensor([b - t, r - l])                                   # crop_bgr.shape[:-1] (refine_css_demo.py:90)
    crop_size, intrinsics, _off = rtools.adjust_intrinsics_crop(np.asarray(K_full, np.float32), crop_size, bbox, max_crop_area)
    return crop_size, intrinsics.float(), bbox


def full_render_case(dec, D, H, W, K, latent, yaw0, trans0):
    """the optimizer's graph (optimizer.py:79-123) with every output enabled and the deterministic functional of G10 as the loss.

    Two backward passes.  (1) the functional over ALL pixels (g_yaw / g_trans / g_latent / g_pcd): each (pixel, surfel) pair contributes O(1-10)
    to these sums with heavy cancellation, so ONE pair flipping across the disc edge -- which a 1e-7 change of a surfel does -- moves them by
    ~4e-3 relative (measured on case a).  (2) the same functional with zero weight on the pixels that hold a pair within 1e-5 of a selection
    threshold (`near_threshold`, stored): r_g_* -- the reference's gradient wherever it is a continuous function of its inputs."""
    grid = ref_grid.Grid3D(D, "cpu", torch.float32)
    lat = torch.tensor(list(latent), dtype=torch.float32, requires_grad=True)
    yaw = torch.tensor([yaw0], requires_grad=True)
    trans = torch.tensor(list(trans0), requires_grad=True)
    renderer = Rasterer(K, (W, H), precision=torch.float32)
    lat_ = F.normalize(lat, p=2, dim=0)
    inputs = torch.cat([lat_.expand(grid.points.size(0), -1), grid.points], 1)
    sdf, _ = dec(inputs)
    pcd, _, normals = grid.get_surface_points(sdf)
    lat.grad = None
    dec.zero_grad()
    grid.points.grad = None
    pcd.retain_grad()
    pose = build_pose(yaw, trans)
    rendering, points = renderer(pcd, normals, normals, pose, primitives="disc", rot="dcm", bg=None, output_depth=True,
                                 output_normals=True, output_nocs=True, output_points=True, output_mask=True)
    salts = {"color": 1, "mask": 2, "depth": 3, "normals": 4, "xyzf": 5}
    near = near_threshold_pixels(K, H, W, pose.detach().numpy(), pcd.detach().numpy(), normals.detach().numpy())
    keep = torch.from_numpy((~near).astype(np.float32)).view(1, H, W)
    Wk = {k: torch.from_numpy(pattern_weights(tuple(rendering[k].shape), salts[k])) for k in rendering}
    lx = (points["xyzf"] * torch.from_numpy(pattern_weights(tuple(points["xyzf"].shape), salts["xyzf"]))).sum()
    loss = sum((rendering[k] * Wk[k]).sum() for k in rendering) + lx
    loss_r = sum((rendering[k] * Wk[k] * keep).sum() for k in rendering) + lx
    loss.backward(retain_graph=True)
    s = sdf.detach().numpy()
    arrs = dict(cfg=np.array([D, H, W]), latent=np.asarray(list(latent), np.float32), yaw=np.asarray([yaw0], np.float32),
                trans=np.asarray(list(trans0), np.float32), K=K.numpy(), pose=pose.detach().numpy(),
                sdf_stride7=s[::7, 0], band_idx=np.nonzero(np.abs(s[:, 0]) < 0.03)[0].astype(np.int32),
                band_margin=np.min(np.abs(np.abs(s[:, 0]) - 0.03)), pcd=pcd.detach().numpy(), normals=normals.detach().numpy(),
                xyzf=points["xyzf"].detach().numpy(), loss=loss.detach().numpy(), g_yaw=yaw.grad.numpy().copy(), g_trans=trans.grad.numpy().copy(),
                g_latent=lat.grad.numpy().copy(), g_pcd=pcd.grad.numpy().copy())
    for t in (yaw, trans, lat, pcd, grid.points):
        t.grad = None
    dec.zero_grad()
    loss_r.backward()
    arrs.update(r_loss=loss_r.detach().numpy(), r_g_yaw=yaw.grad.numpy().copy(), r_g_trans=trans.grad.numpy().copy(),
                r_g_latent=lat.grad.numpy().copy(), r_g_pcd=pcd.grad.numpy().copy())
    dot = ((normals.detach() @ pose.detach()[:3, :3].t()) * points["xyz"].detach()).sum(1)
    arrs["filt_margin"] = dot.abs().min().numpy()
    for k, v in rendering.items():
        arrs["out_" + k] = v.detach().numpy()
    arrs["near_threshold"] = np.packbits(near)
    cov = int((rendering["mask"].detach() > 0).sum())
    print("   N", pcd.shape[0], "Nf", points["xyzf"].shape[0], "covered px", cov, "of", H * W, "loss", float(loss), "g_yaw", arrs["g_yaw"],
          "g_trans", arrs["g_trans"], "g_lat", arrs["g_latent"], "| robust:", arrs["r_g_yaw"], arrs["r_g_trans"], arrs["r_g_latent"],
          "band margin", arrs["band_margin"], "filt margin", arrs["filt_margin"], "near px", int(near.sum()))
    return arrs


# renderer-space poses (metric / scale with scale 2, refine_css_demo.py:159): lateral offset +-2 (= +-4 m), depth 4 ... 12.5 (= 8 ... 25 m),
# camera 0.4 above the object centre
G14_CASES = {
    "a": dict(K_full=KITTI_K, yaw=0.9, trans=(2.0, 0.42, 6.0), latent=(0.5, -0.3, 0.6)),           # 12 m, 4 m to the right
    "b": dict(K_full=KITTI_K_ANISO, yaw=-0.7, trans=(-2.0, 0.40, 12.5), latent=(-0.6, 0.2, 0.1)),  # 25 m, 4 m to the left, fx != fy
    "c": dict(K_full=KITTI_K, yaw=2.4, trans=(2.0, 0.45, 4.0), latent=(0.3, -0.5, 0.8)),           # 8 m: the box leaves the frame's right edge
}


def g14():
    """(a) Rasterer.forward + autograd gradients at the configs[1..3] crop area (~256x192 rays) with crop intrinsics produced by the reference's
    own adjust_intrinsics_crop for KITTI-like boxes; three objects 8 - 25 m away and 4 m off the optical axis."""
    dec = load_fitted()[0]
    arrs = {}
    for tag, c in G14_CASES.items():
        (H, W), K, bbox = kitti_crop_intrinsics(c["K_full"], c["yaw"], list(c["trans"]), 256 * 192)
        print("G14", tag, "bbox", bbox, "crop HxW", H, W, "K", K.numpy().round(2).tolist())
        r = full_render_case(dec, 40, H, W, K, c["latent"], c["yaw"], c["trans"])
        r["bbox"] = np.asarray(bbox, np.int32)
        arrs.update({tag + "_" + k: v for k, v in r.items()})
    save("g14_cropped_intrinsics.npz", **arrs)


def g14o():
    """(b) the reference Optimizer's 10-iteration trajectory in its real regime: rendering_area = 32 (config_refine.ini:12 -> ~1024 rays per
    crop) and the cropped, off-centre intrinsics of case a; D = 40."""
    from pipelines.optimizer import Optimizer
    dec = load_fitted()[0]
    arrs = {}
    for tag, D in (("a", 40), ("c", 20)):
        c = G14_CASES[tag]
        (H, W), K, bbox = kitti_crop_intrinsics(c["K_full"], c["yaw"], list(c["trans"]), 32 * 32)
        yaw_gt, trans_gt, lat_gt = c["yaw"], list(c["trans"]), [0.3, -0.5, 0.8]
        # targets rendered from the ground-truth pose with the reference renderer (as synth_targets, with the crop's K)
        grid = ref_grid.Grid3D(D, "cpu", torch.float32)
        lat_ = F.normalize(torch.tensor(lat_gt), p=2, dim=0)
        sdf, _ = dec(torch.cat([lat_.expand(grid.points.size(0), -1), grid.points], 1))
        pcd, _, normals = grid.get_surface_points(sdf)
        renderer = Rasterer(K, (W, H), precision=torch.float32)
        rendering, points = renderer(pcd.detach(), normals.detach(), normals.detach(), build_pose(torch.tensor([yaw_gt]), torch.tensor(trans_gt)),
                                     primitives="disc", rot="dcm", output_nocs=True, output_points=True, output_mask=True)
        nocs = rendering["color"].detach()
        lidar = (points["xyzf"].detach() * 2.0)[::3].numpy().copy()
        params = {"yaw": [yaw_gt + 0.12], "trans": [trans_gt[0] + 0.05, trans_gt[1] + 0.02, trans_gt[2] - 0.08], "scale": [2.0],
                  "latent": [0.5, -0.3, 0.6]}
        opt = Optimizer({k: list(v) for k, v in params.items()}, "cpu", {"2d": 0.3, "3d": 0.5})
        grid = ref_grid.Grid3D(D, "cpu", torch.float32)
        traj, buf = [], io.StringIO()
        for it in range(10):
            with contextlib.redirect_stdout(buf):
                opt.optimize(1, nocs, lidar, dec, grid, K, [H, W], viz_type=None)
            traj.append(np.concatenate([opt.params[k].detach().numpy().ravel() for k in ("yaw", "trans", "scale", "latent")]))
        l2d, l3d = [], []
        for l in (l for l in buf.getvalue().splitlines() if l.startswith("ITER")):
            parts = l.split("2D - ")[1].split(", 3D - ")
            l2d.append(float(parts[0])); l3d.append(float(parts[1].split(", Total")[0]))
        print("G14o", tag, "crop HxW", H, W, "K", K.numpy().round(2).tolist(), "covered px", int((rendering["mask"] > 0).sum()), "lidar", lidar.shape[0],
              "yaw", [round(float(t[0]), 4) for t in traj], "l2d", l2d[0], l2d[-1], "l3d", l3d[0], l3d[-1])
        assert len(l2d) == 10
        arrs.update({tag + "_" + k: v for k, v in dict(
            D=D, H=H, W=W, K=K.numpy(), bbox=np.asarray(bbox, np.int32), nocs_target=nocs.numpy(), lidar=lidar,
            init=np.concatenate([np.asarray(params[k], np.float32) for k in ("yaw", "trans", "scale", "latent")]),
            traj=np.asarray(traj), loss2d_weighted=np.asarray(l2d), loss3d_weighted=np.asarray(l3d)).items()})
    save("g14o_optimizer_cropped.npz", **arrs)


def load_asset(name, precision=torch.float32):
    return ref_ws.setup_dsdf(os.path.join(HERE, "..", "sdflabel_amd", "assets", name + ".pt"), precision=precision)[0]


def g14e():
    """(c) a SECOND decoder at full size: the ellipsoid fit (tools/fit_decoder.py --shape ellipsoid; smooth normals, ~0.7 k band surfels at
    D = 40) -- one centred 256x256 crop as G10 and one crop with the cropped intrinsics of case a -- and its LayerNorm variant
    (weight_norm=False, deep_sdf_decoder_scale.py:56-57,99-101) on the centred crop."""
    arrs = {}
    c = G14_CASES["a"]
    (Hc, Wc), Kc, bbox = kitti_crop_intrinsics(c["K_full"], c["yaw"], list(c["trans"]), 256 * 192, half_extents=(0.6, 0.45, 0.92))
    for tag, asset, H, W, K, yaw0, trans0 in (("wn_centre", "deepsdf_synth_ellipsoid", 256, 256, K_for(256, 256), 0.7, (0.03, 0.02, 3.45)),
                                              ("wn_crop", "deepsdf_synth_ellipsoid", Hc, Wc, Kc, c["yaw"], c["trans"]),
                                              ("ln_centre", "deepsdf_synth_ellipsoid_ln", 256, 256, K_for(256, 256), -1.1, (0.2, -0.1, 2.9))):
        if not os.path.isfile(os.path.join(HERE, "..", "sdflabel_amd", "assets", asset + ".pt")):
            print("G14e: asset", asset, "missing, skipped")
            continue
        print("G14e", tag, "crop HxW", H, W)
        r = full_render_case(load_asset(asset), 40, H, W, K, (0.5, -0.3, 0.6), yaw0, trans0)
        arrs.update({tag + "_" + k: v for k, v in r.items()})
    save("g14e_second_decoder.npz", **arrs)


def _ref_grads_precision(prec, D, H, W, r0, r1, latent, yaw0, trans0):
    """the reference's autograd gradients (yaw, trans, latent) of two functionals of the rendering of image rows [r0, r1) at `prec`:
    'sum' = color.sum() + mask.sum() + normals.sum() + xyzf.sum() (what bench.py back-propagates) and 'pat' = the hash-weighted functional of
    G10.  Rows are selected by the camera, not by slicing: a Rasterer of r1 - r0 rows whose principal point is shifted up by r0 sees exactly the
    rays of those rows (K' = K with cy - r0; exact in float16 for the values used) -- the dense N x P tensors of a full 512x512 float32 backward
    (~60 GB) do not fit this machine, four 128-row strips do.  The functionals are sums over pixels, so strip gradients add up."""
    dec, _ = load_fitted(prec)
    grid = ref_grid.Grid3D(D, "cpu", prec)
    lat = torch.tensor(latent, requires_grad=True)
    yaw = torch.tensor([yaw0], requires_grad=True)
    trans = torch.tensor(trans0, requires_grad=True)
    lat_ = F.normalize(lat.to(prec), p=2, dim=0)                                                     # optimizer.py:96
    inputs = torch.cat([lat_.expand(grid.points.size(0), -1), grid.points], 1).to(lat_.dtype)       # :99-100
    sdf, _ = dec(inputs)
    pcd, _, normals = grid.get_surface_points(sdf)
    lat.grad = None
    dec.zero_grad()
    grid.points.grad = None
    pose = torch.eye(4).to(prec)                                                                      # :86-90
    pose[:3, :3] = rtools.rot_from_yaw(yaw).to(prec)
    pose[1] *= -1
    pose[:3, 3] = trans.to(prec)
    K = K_for(H, W)
    K[1, 2] -= r0
    r = Rasterer(K.to(prec), (W, r1 - r0), precision=prec)
    rend, pts = r(pcd, normals, normals, pose, primitives="disc", rot="dcm", bg=None, output_depth=False, output_normals=True, output_nocs=True,
                  output_points=True, output_mask=True)
    first = r0 == 0                                                                                   # the xyzf term belongs to the image once
    salts = {"color": 1, "mask": 2, "normals": 4, "xyzf": 5}
    out = {}
    for name in ("sum", "pat"):
        loss = 0
        for k in ("color", "mask", "normals"):
            if name == "sum":
                loss = loss + rend[k].float().sum()
            else:
                wk = pattern_weights((rend[k].shape[0], H, W), salts[k])[:, r0:r1]
                loss = loss + (rend[k].float() * torch.from_numpy(np.ascontiguousarray(wk))).sum()
        if first:
            x = pts["xyzf"].float()
            loss = loss + (x.sum() if name == "sum" else (x * torch.from_numpy(pattern_weights(tuple(x.shape), salts["xyzf"]))).sum())
        for t in (yaw, trans, lat):
            t.grad = None
        loss.backward(retain_graph=(name == "sum"))
        out[name] = np.concatenate([yaw.grad.numpy().ravel(), trans.grad.numpy().ravel(), lat.grad.numpy().ravel()]).astype(np.float64)
        out[name + "_loss"] = float(loss)
    return out, int(pcd.shape[0])


def g11g():
    """BASELINE configs[4], gradients: the reference's OWN float16 autograd gradients w.r.t. yaw, trans and latent at 512x512, D = 40 (the
    whole image in one pass) beside its float32 gradients (four 128-row strips, see _ref_grads_precision), for two functionals.  The gap
    between the two precisions is the yardstick of the tolerance stated in tests/test_gpu_configs.py."""
    D, H, W = 40, 512, 512
    latent, yaw0, trans0 = [0.3, -0.5, 0.8], 0.6, [0.0, 0.0, 3.5]
    arrs = dict(cfg=np.array([D, H, W]), latent=np.asarray(latent, np.float32), yaw=np.asarray([yaw0], np.float32),
                trans=np.asarray(trans0, np.float32), K=K_for(H, W).numpy())
    g16, n16 = _ref_grads_precision(torch.float16, D, H, W, 0, H, latent, yaw0, trans0)
    print("G11g f16", n16, g16)
    g32 = None
    for r0 in range(0, H, 128):
        g, n32 = _ref_grads_precision(torch.float32, D, H, W, r0, r0 + 128, latent, yaw0, trans0)
        g32 = g if g32 is None else {k: g32[k] + g[k] for k in g}
        print("G11g f32 strip", r0, {k: v for k, v in g.items() if k.endswith("loss")})
    # cross-check of the strip decomposition in float16, where the whole image fits: strips must add up to the one-pass result up to float16 noise
    print("G11g f32", n32, g32)
    for name in ("sum", "pat"):
        arrs["f16_g_" + name] = g16[name]
        arrs["f32_g_" + name] = g32[name]
        arrs["f16_loss_" + name] = g16[name + "_loss"]
        arrs["f32_loss_" + name] = g32[name + "_loss"]
        print("G11g", name, "gap |f16 - f32|", np.abs(g16[name] - g32[name]), "scale", np.abs(g32[name]).max())
    arrs["f16_n_surfels"], arrs["f32_n_surfels"] = n16, n32
    save("g11g_config4_fp16_grads.npz", **arrs)


def g14p():
    """the secondary configurations in the cropped camera regime: primitives 'circle' (with and without a background image) and the quaternion
    pose path with the disc primitive, crop intrinsics of case a at ~48x64 rays, D = 20 surfels: images + autograd gradients w.r.t. the surfel
    positions and the pose (the weights of the functional vanish on pixels that hold a pair within 1e-5 of a disc-edge / |n.ray| threshold).
    ('circle_opt' is not part of it: with these intrinsics the reference itself raises -- "numel: integer multiplication overflow" from the
    torch.sparse.FloatTensor it builds at primitives.py:135 -- so there is nothing to pin.)"""
    dec = load_fitted()[0]
    c = G14_CASES["a"]
    (H, W), K, bbox = kitti_crop_intrinsics(c["K_full"], c["yaw"], list(c["trans"]), 48 * 64)
    _, grid, lat, sdf, pts, nocs, nrm = surface_case(20, [0.5, -0.3, 0.6], dec)
    pts0, nrm0 = pts.detach(), nrm.detach()
    arrs = dict(cfg=np.array([20, H, W]), K=K.numpy(), points=pts0.numpy(), normals=nrm0.numpy(), yaw=np.asarray([c["yaw"]], np.float32),
                trans=np.asarray(c["trans"], np.float32))
    r = Rasterer(K, (W, H), precision=torch.float32)
    gen = torch.Generator().manual_seed(23)
    bgimg = torch.rand(3, H, W, generator=gen)
    arrs["bg"] = bgimg.numpy()
    pose0 = build_pose(torch.tensor([c["yaw"]]), torch.tensor(list(c["trans"])))
    near = near_threshold_pixels(K, H, W, pose0.numpy(), pts0.numpy(), nrm0.numpy())
    keep = torch.from_numpy((~near).astype(np.float32)).view(1, H, W)
    arrs["near_threshold"] = np.packbits(near)
    for tag, prim, use_bg, rot in (("circle_bg0", "circle", False, "dcm"), ("circle_bg1", "circle", True, "dcm"), ("disc_quat", "disc", False, "quat")):
        p = pts0.clone().requires_grad_(True)
        if rot == "dcm":
            yaw = torch.tensor([c["yaw"]], requires_grad=True)
            trans = torch.tensor(list(c["trans"]), requires_grad=True)
            cam = build_pose(yaw, trans)
            leaves = {"g_yaw": yaw, "g_trans": trans}
        else:
            # the same rigid motion as a quaternion + translation WITHOUT the row-1 flip of the optimizer (the quat path has none,
            # projection.py:104-199): rotate about y by yaw, then translate
            half = c["yaw"] / 2.0
            cam = torch.tensor([np.cos(half), 0.0, np.sin(half), 0.0] + list(c["trans"]), dtype=torch.float32, requires_grad=True)
            leaves = {"g_cam": cam}
        rend = r(p, nrm0, nrm0, cam, rot=rot, primitives=prim, bg=bgimg if use_bg else None, output_mask=True,
                 output_depth=not use_bg, output_normals=not use_bg, output_nocs=True, output_points=False)
        Ws = {k: torch.randn(v.shape, generator=gen) * (keep if prim == "disc" else 1.0) for k, v in rend.items()}
        loss = sum((rend[k] * Ws[k]).sum() for k in rend)
        loss.backward()
        for k, v in rend.items():
            arrs[tag + "_out_" + k] = v.detach().numpy()
            arrs[tag + "_W_" + k] = Ws[k].numpy()
        arrs[tag + "_cam"] = cam.detach().numpy()
        arrs[tag + "_g_points"] = p.grad.numpy()
        for k, t in leaves.items():
            arrs[tag + "_" + k] = t.grad.numpy()
        print("G14p", tag, "crop HxW", H, W, "covered px", int((rend["mask"] > 0).sum()), "loss", float(loss), {k: t.grad.numpy().round(3).tolist() for k, t in leaves.items()},
              "|g_points|max", float(p.grad.abs().max()))
    save("g14p_secondary_cropped.npz", **arrs)


ALL = {"G1": g1, "G2": g2, "G3": g3, "G4": g4, "G5": g5, "G6": g6, "G7": g7, "G8": g8, "G8b": g8b, "G8c": g8c, "G8h": g8h, "G8s": g8s, "G9": g9, "G10": g10, "G10b": g10b, "G11": g11, "G12": g12, "G13": g13, "G13s": g13s, "G11g": g11g, "G14": g14, "G14p": g14p, "G14o": g14o, "G14e": g14e}

if __name__ == "__main__":
    which = sys.argv[1:] or list(ALL)
    for w in which:
        ALL[w]()
