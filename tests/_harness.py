"""Test harness: the reference's refinement loop (pipelines/optimizer.py:26-54, 79-237) restated on top of the drop-in modules.

The reference's `Optimizer` class cannot travel to the GPU box, so its loop is restated here in this project's own words for the
tests only (SURVEY.md §8 a8 / a-harness): same parameter groups and solvers (Adam lr .01 on yaw and trans; SGD lr .01 on scale,
3e-5 on latent, :34-52), same per-iteration order of operations, same 3-D nearest-neighbour loss (exact NN, sklearn KDTree on the
host, :166-198) and 2-D NOCS window loss (:200-237).  The renderer calls go through sdflabel_amd exactly as optimizer.py issues them.
"""
import numpy as np
import torch
import torch.nn.functional as F
from sklearn.neighbors import KDTree


def rot_from_yaw(yaw):
    c, s = torch.cos(yaw), torch.sin(yaw)
    z, o = yaw.new_zeros(1), yaw.new_ones(1)
    return torch.stack((c, z, s, z, o, z, -s, z, c)).view(3, 3)


def loss_3d(pcd_est, pcd_lidar, scale, threshold=0.2):
    """mean distance of the estimated points to their nearest lidar point, pairs closer than threshold/scale only"""
    if pcd_est.nelement() == 0 or pcd_lidar.nelement() == 0:
        return pcd_est.new_zeros(())
    tree = KDTree(pcd_lidar.detach().cpu().numpy())
    dists, idxs = tree.query(pcd_est.detach().cpu().numpy())
    dists, idxs = dists[:, 0], idxs[:, 0]
    close = torch.from_numpy(dists < threshold / float(scale)).to(pcd_est.device)
    idxs = torch.from_numpy(idxs).to(pcd_est.device)
    d = (pcd_lidar[idxs[close]] - pcd_est[close]).norm(p=2, dim=1)
    return d.mean() if d.nelement() else pcd_est.new_zeros(())


def loss_2d(rendering_nocs, css_nocs, diam=5, threshold_nocs=1):
    """for every rendered (non-zero) pixel: smallest NOCS distance to the target inside a soft window of radius `diam` around it
    (target weighted by clamp(diam - pixel distance, 0)), mean over the pixels whose minimum is below threshold_nocs"""
    nz = rendering_nocs.sum(0).nonzero()
    if not int(nz.sum()):
        return rendering_nocs.new_zeros(())
    Hh, Ww = rendering_nocs.shape[1:]
    xx, yy = torch.meshgrid(torch.arange(Hh), torch.arange(Ww), indexing="ij")
    grid = torch.stack((xx, yy), -1).float().to(rendering_nocs.device).view(1, -1, 2)
    dist = torch.clamp(diam - (grid - nz.view(-1, 1, 2).float()).pow(2).sum(-1).sqrt(), min=0).view(-1, 1, Hh, Ww)
    masked = css_nocs.unsqueeze(0) * dist
    vals = rendering_nocs[:, nz[:, 0], nz[:, 1]].t()
    diff = (masked - vals.unsqueeze(-1).unsqueeze(-1)).pow(2).sum(1).sqrt()
    dmin = diff.view(diff.shape[0], -1).min(1)[0]
    return dmin[dmin < threshold_nocs].mean()


class Refiner:
    def __init__(self, params, device, weights):
        self.p = {k: torch.tensor(v, dtype=torch.float32, device=device, requires_grad=True) for k, v in params.items()}
        self.adam = torch.optim.Adam([{"params": self.p["yaw"], "lr": 0.01}, {"params": self.p["trans"], "lr": 0.01}], lr=0.03)
        self.sgd = torch.optim.SGD([{"params": self.p["scale"], "lr": 0.01}, {"params": self.p["latent"], "lr": 0.00003}], lr=0.01,
                                   momentum=0.0)
        self.weights = weights
        self.log = []

    def zero(self):
        self.adam.zero_grad()
        self.sgd.zero_grad()

    def optimize(self, iters, nocs_pred, lidar_np, dsdf, grid, renderer):
        dev = grid.points.device
        for _ in range(iters):
            self.zero()
            lidar = (torch.tensor(lidar_np, dtype=torch.float32, device=dev) / self.p["scale"])
            pose = torch.eye(4, device=dev)
            pose[:3, :3] = rot_from_yaw(self.p["yaw"])
            pose[1] *= -1
            pose[:3, 3] = self.p["trans"]
            latent_ = F.normalize(self.p["latent"], p=2, dim=0)
            inputs = torch.cat([latent_.expand(grid.points.size(0), -1), grid.points], 1)
            sdf, _ = dsdf(inputs)
            pcd, _, normals = grid.get_surface_points(sdf)
            self.zero()
            rendering, points = renderer(pcd, normals, normals, pose, primitives='disc', rot='dcm', bg=None, output_depth=False,
                                         output_normals=True, output_nocs=True, output_points=True, output_mask=True)
            est = points['xyzf']
            if est.nelement() == 0 or lidar.nelement() == 0:
                continue
            l3 = loss_3d(est, lidar, self.p["scale"][0].item())
            target = F.interpolate(nocs_pred.unsqueeze(0), size=rendering['color'].shape[1:], mode='nearest').squeeze(0)
            l2 = loss_2d(rendering['color'], target)
            loss = self.weights['3d'] * l3 + self.weights['2d'] * l2
            if torch.isnan(loss).sum() > 0 or loss.sum() == 0:
                continue
            self.log.append((self.weights['2d'] * l2.item(), self.weights['3d'] * l3.item()))
            loss.backward()
            self.adam.step()
            self.sgd.step()

    def vector(self):
        return np.concatenate([self.p[k].detach().cpu().numpy().ravel() for k in ("yaw", "trans", "scale", "latent")])
