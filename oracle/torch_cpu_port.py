"""Multi-threaded CPU port of the reference's renderer hot path in torch (CPU) ops.  TEST / BASELINE INFRASTRUCTURE ONLY.

What it is for: bench.py's `cpu_baseline` leg -- "the reference CPU renderer timed on the node's host cores" (BASELINE.json north_star).
The reference itself cannot travel to the GPU box, and the numpy oracle (oracle/sdf_oracle.py) runs its elementwise passes on one core;
this module restates the SAME dense algorithm with torch CPU tensors and autograd, operation by operation in the reference's order, so
that it costs what the reference costs on the same cores: the decoder as nn-style linears whose weight-norm is re-evaluated on every call
and whose parameters require grad (the reference never freezes them: the backward also produces the unneeded weight gradients,
SURVEY.md §8 a3), normals through an autograd backward of sdf.sum() (grid.py:55), dense N x P splat tensors, autograd backward.
Each function cites the reference file:line it follows (paths relative to /root/reference).

Only bench.py's cpu_baseline and tests/ may import it; the product never does.  Pinned by tests/test_oracle_golden.py against the
golden G7 (images + autograd gradients captured from the reference) -- it must give the reference's numbers, not just its cost.
"""
import numpy as np
import torch
import torch.nn.functional as F


class DecoderPort(torch.nn.Module):
    """deep_sdf_decoder_scale.py:9-114 for the weight-norm decoder family (norm_layers with weight_norm=True), eval mode."""

    def __init__(self, state, spec):
        super().__init__()
        self.latent_in = list(spec.get("latent_in", ()))
        self.n_lin = len(spec["dims"]) + 1
        self.use_tanh = bool(spec.get("use_tanh", False))
        for l in range(self.n_lin):
            for key in ("weight_v", "weight_g", "weight", "bias"):
                k = "lin%d.%s" % (l, key)
                if k in state:
                    self.register_parameter("l%d_%s" % (l, key), torch.nn.Parameter(torch.as_tensor(np.asarray(state[k], np.float32))))

    def forward(self, inp):
        x = inp
        for l in range(self.n_lin):                                                       # :88
            if l in self.latent_in:
                x = torch.cat([x, inp], 1)                                                # :90-91
            v = getattr(self, "l%d_weight_v" % l, None)
            if v is not None:                                                             # nn.utils.weight_norm: w = g * v / ||v||, every call (:51-54)
                w = v * (getattr(self, "l%d_weight_g" % l) / v.norm(2, dim=1, keepdim=True))
            else:
                w = getattr(self, "l%d_weight" % l)
            x = F.linear(x, w, getattr(self, "l%d_bias" % l))                             # :94
            if l == self.n_lin - 1 and self.use_tanh:
                x = torch.tanh(x)                                                         # :96-97
            if l < self.n_lin - 1:
                x = F.relu(x)                                                             # :102
        return torch.tanh(x)                                                              # :106-107


def generate_point_grid(D):
    """grid.py:22-41"""
    lin = np.mgrid[-1:1:D * 1j]
    X, Y, Z = np.meshgrid(lin, lin, lin, indexing="ij")
    g = np.stack([X, Y, Z], axis=-1).reshape(-1, 3)
    g[1::2, :2] += (lin.max() - lin.min()) / D / 2
    return torch.from_numpy(g.astype(np.float32))


def get_surface_points(points, sdf, threshold=0.03):
    """grid.py:43-71: normals = d sum(sdf) / d points by an autograd backward over the whole graph (retain_graph), in-place normalisation
    (norm detached), projection x - sdf * n_hat, band |sdf| < threshold, order-preserving compaction."""
    sdf.sum().backward(retain_graph=True)                                                # :55 (fills every leaf's .grad, decoder parameters included)
    n = points.grad.detach().clone()                                                     # the hook's copy, :11-12
    n = n / n.norm(2, dim=1, keepdim=True)                                               # :57-58
    p = points - sdf * n                                                                 # :61
    band = (sdf.abs() < threshold).expand_as(p)                                          # :64
    pm = p.masked_select(band).view(-1, 3)                                               # :65
    nm = n.masked_select(band).view(-1, 3)                                               # :66
    return pm, (pm + 1) / 2, nm                                                          # :67-71


def project_in_2D(K, pose, points, normals, res):
    """projection.py:7-101 (dcm, NOCS colours, filter_normals)."""
    eps = torch.finfo(K.dtype).eps
    RT = pose[:-1, :]                                                                    # :34
    ch = torch.cat([points, torch.ones_like(points[:, :1])], -1).t()                     # :44-46
    n_p = (RT[:, :3] @ normals.t()).t()                                                  # :49
    col = points.clone()
    col[:, 0] *= -1                                                                      # :53-55
    p3 = (RT @ ch).t()                                                                   # :58
    dot = torch.bmm(n_p.unsqueeze(-2), p3.unsqueeze(-1)).squeeze(-1)                     # :62
    keep = dot < 0
    out = {"points_3d_filt": p3.masked_select(keep).view(-1, 3), "colors_3d_filt": col.masked_select(keep).view(-1, 3)}   # :64-66
    h = (K @ p3.t()).t()                                                                 # :88
    uv = h[:, :2] / (h[:, 2:] + eps)                                                     # :89
    out.update(points_3d=p3, normals_3d=n_p, colors_3d=col,
               points_2d=torch.cat([uv[:, 0:1].clamp(-1, res[0]), uv[:, 1:2].clamp(-1, res[1])], -1))
    return out


def inside_surfel(K, grid_2d, vertex_3d, normals, diam=0.04, depth_constant=150):
    """primitives.py:165-242 with softclamp=False, add_bg=False: the dense (N, P) weights."""
    dt = K.dtype
    eps = torch.finfo(dt).eps
    n_v3d = (normals * vertex_3d).sum(1)                                                 # :202
    g = torch.cat([grid_2d[0].to(dt), torch.ones_like(grid_2d[0][:, :1]).to(dt)], -1)    # :203-207
    rays = (K.float().inverse() @ g.t()).t()                                             # :204-208
    b = (normals @ rays.t())                                                             # :209  (N, P)
    b[b.abs() < 0.01] = eps                                                              # :210 (in place: no gradient through the overwritten entries)
    z = n_v3d.unsqueeze(-1) / b                                                          # :211
    g3 = rays.unsqueeze(0) * z.unsqueeze(-1)                                             # :212  (N, P, 3)
    d = (vertex_3d.view(-1, 1, 3) - g3).pow(2).sum(-1).sqrt()                            # :215,:220
    dist = torch.clamp(diam - d, min=0)                                                  # :220
    mask = (dist > 0).detach().to(dt)                                                    # :226
    zz = -z * mask                                                                       # :227
    zn = torch.norm(zz, p=2, dim=0).detach().unsqueeze(0)                                # :228
    zz = torch.clamp(zz / (zn + eps) + 1, min=0) * depth_constant                        # :229-230
    zz = zz.masked_fill(mask == 0, torch.finfo(dt).min)                                  # :240
    return F.softmax(zz, dim=0) * mask                                                   # :240


def rasterer_forward(K, res, coords, normals, pose, output_depth=False):
    """rasterer.py:49-155 with primitives='disc', rot='dcm', output_nocs/mask/normals/points as the optimizer asks (optimizer.py:110-123)."""
    W, H = res
    yy, xx = np.mgrid[0:H, 0:W]
    grid = torch.from_numpy(np.stack((xx, yy), axis=-1).reshape((1, -1, 2)))             # :25-27
    proj = project_in_2D(K, pose, coords, normals, res)
    v3, nrm, col = proj["points_3d"], proj["normals_3d"], proj["colors_3d"]
    prob = inside_surfel(K, grid, v3, nrm).unsqueeze(1).expand(-1, 3, -1)                # :101-104, primitives.py:241
    out = {"color": torch.clamp((prob * ((col + 1) / 2).unsqueeze(-1)).sum(0), max=1).view(3, H, W),        # :113-124
           "mask": torch.clamp(prob[:, :1].sum(0), max=1).view(1, H, W),                                     # :127-131
           "normals": torch.clamp((prob * ((nrm + 1) / 2).unsqueeze(-1)).sum(0), max=1).view(3, H, W)}       # :140-144
    if output_depth:
        out["depth"] = (prob[:, :1] * v3[:, 2:3].unsqueeze(-1)).sum(0).view(1, H, W)                        # :134-137
    points = {"xyzf": proj["points_3d_filt"], "rgbf": (proj["colors_3d_filt"] + 1) / 2, "xyz": v3, "rgb": (col + 1) / 2}   # :147-153
    return out, points


def build_pose(yaw, trans):
    """optimizer.py:86-90 (utils/refinement.py:108-125)"""
    c, s = torch.cos(yaw), torch.sin(yaw)
    z, o = torch.zeros_like(c), torch.ones_like(c)
    pose = torch.eye(4)
    pose[:3, :3] = torch.stack((c, z, s, z, o, z, -s, z, c)).view(3, 3)
    pose[1] *= -1
    pose[:3, 3] = trans
    return pose


def crop_iteration(decoder, grid_points, K, res, yaw, trans, latent, weights=None, output_depth=False):
    """One crop-iteration of the refinement loop without the losses (optimizer.py:79-123,156): decoder on the grid, surface extraction,
    rendering, backward of a linear functional of the outputs (weights: dict of tensors; default all-ones = plain sums) to yaw, trans,
    latent.  Returns (rendering, points, number of surfels)."""
    for p in list(decoder.parameters()) + [grid_points, yaw, trans, latent]:
        p.grad = None
    latent_ = F.normalize(latent, p=2, dim=0)                                            # optimizer.py:96
    inputs = torch.cat([latent_.expand(grid_points.size(0), -1), grid_points], 1)        # :99-100
    sdf = decoder(inputs)                                                                # :101
    pcd, _, normals = get_surface_points(grid_points, sdf)                               # :104
    for p in list(decoder.parameters()) + [grid_points, yaw, trans, latent]:             # :107 (solver.zero_grad)
        p.grad = None
    pose = build_pose(yaw, trans)
    rendering, points = rasterer_forward(K, res, pcd, normals, pose, output_depth=output_depth)     # :110-123
    loss = 0
    for k, v in rendering.items():
        loss = loss + ((v * weights[k]).sum() if weights is not None else v.sum())
    loss = loss + ((points["xyzf"] * weights["xyzf"]).sum() if weights is not None else points["xyzf"].sum())
    if weights is not None:
        for k in ("rgbf", "xyz", "rgb"):
            if k in weights:
                loss = loss + (points[k] * weights[k]).sum()
    loss.backward()                                                                      # :156
    return rendering, points, pcd.shape[0], loss.detach()
