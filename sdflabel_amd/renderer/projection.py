"""project_in_2D / project_in_2D_quat -- the reference's standalone projection functions (sdfrenderer/renderer/projection.py:7-199) on top
of the HIP projection kernels (csrc/project.hip), same signatures and output dictionaries.

`Rasterer.forward` does not go through these (it fuses projection and splat behind one autograd node); they exist for callers that use the
reference's `renderer.projection` module directly.  The HPR (hidden point removal) filter branch (projection.py:72-85) is a scipy convex-hull
pass on the host in the reference and is outside the renderer hot path: filter_hpr=True raises.
"""
import torch

from .. import _lib
from .utils_rasterer import qrot_matrix


class _ProjectFn(torch.autograd.Function):
    """(points, normals, colors|None, pose44) -> p_cam, n_cam, col   (+ front-facing index list through `holder`)"""

    @staticmethod
    def forward(ctx, points, normals, colors, pose, K, res, nocs_mode, want_filter, holder):
        L = _lib.lib()
        dev = points.device
        n = points.shape[0]
        m = max(n, 1)
        f32 = dict(dtype=torch.float32, device=dev)
        pts_c, nrm_c = points.detach().contiguous().float(), normals.detach().contiguous().float()
        col_c = None if nocs_mode else colors.detach().contiguous().float()
        pose_c = pose.detach().contiguous().float()
        p_cam, n_cam, col = torch.empty((m, 3), **f32), torch.empty((m, 3), **f32), torch.empty((m, 3), **f32)
        fidx = torch.empty((m,), dtype=torch.int32, device=dev) if want_filter else None
        fcnt = torch.zeros((1,), dtype=torch.int32, device=dev) if want_filter else None
        if n > 0:
            with _lib.guard(points):
                _lib.check(L.sdfr_project_dcm(_lib.ptr(pose_c), _lib.ptr(K), _lib.ptr(pts_c), _lib.ptr(nrm_c), _lib.ptr(col_c), 1, n, None,
                                              int(nocs_mode), res[0], res[1], _lib.ptr(p_cam), _lib.ptr(n_cam), _lib.ptr(col), None,
                                              _lib.ptr(fidx), _lib.ptr(fcnt), None, None, _lib.stream_ptr()), "sdfr_project_dcm")
        nf = int(fcnt.item()) if (want_filter and n > 0) else 0
        holder["fidx"] = fidx[:nf].long() if want_filter else None
        ctx.save_for_backward(pts_c, nrm_c, pose_c)
        ctx.cfg = (n, nocs_mode)
        return p_cam[:n], n_cam[:n], col[:n]

    @staticmethod
    def backward(ctx, g_p, g_n, g_c):
        L = _lib.lib()
        pts, nrm, pose = ctx.saved_tensors
        n, nocs_mode = ctx.cfg
        dev = pts.device
        f32 = dict(dtype=torch.float32, device=dev)
        m = max(n, 1)

        def full(g):
            out = torch.zeros((m, 3), **f32)
            if g is not None and n > 0:
                out[:n] = g
            return out

        g_p, g_n, g_c = full(g_p), full(g_n), full(g_c)
        g_points, g_normals = torch.zeros((m, 3), **f32), torch.zeros((m, 3), **f32)
        g_colors = None if nocs_mode else torch.zeros((m, 3), **f32)
        g_pose = torch.zeros((4, 4), **f32)
        if n > 0:
            with _lib.guard(pts):
                _lib.check(L.sdfr_project_dcm_bwd(_lib.ptr(pose), _lib.ptr(pts), _lib.ptr(nrm), _lib.ptr(g_p), _lib.ptr(g_n), _lib.ptr(g_c), 1, n,
                                                  None, int(nocs_mode), _lib.ptr(g_points), _lib.ptr(g_normals), _lib.ptr(g_colors),
                                                  _lib.ptr(g_pose), None, None, _lib.stream_ptr()), "sdfr_project_dcm_bwd")
        return g_points[:n], g_normals[:n], None if nocs_mode else g_colors[:n], g_pose, None, None, None, None, None


def _project(K, pose44, points, normals, colors, resolution_px, filter_normals, filter_hpr, nocs_mode):
    if filter_hpr:
        raise NotImplementedError("filter_hpr (projection.py:72-85) is a host-side scipy convex-hull filter outside the renderer hot path")
    _lib.require_gpu_float(points, normals, None if nocs_mode else colors)
    dev, dtype = points.device, K.dtype                                            # the reference computes in K's dtype (:27-28)
    res_x, res_y = resolution_px
    eps = torch.finfo(dtype).eps
    Kf = K.detach().to(dev, torch.float32).contiguous()
    holder = {}
    p_cam, n_cam, col = _ProjectFn.apply(points, normals, None if nocs_mode else colors, pose44.to(dev, torch.float32), Kf, (int(res_x), int(res_y)),
                                         nocs_mode, bool(filter_normals), holder)
    out = {}
    if filter_normals:                                                             # :61-70
        fidx = holder["fidx"]
        out['points_3d_filt'] = p_cam.index_select(0, fidx).to(dtype)
        out['normals_3d_filt'] = n_cam.index_select(0, fidx).to(dtype)
        out['colors_3d_filt'] = col.index_select(0, fidx).to(dtype)
    # pixel projection (:88-93) with torch ops on the kernel's camera-frame points, so that points_2d is differentiable as in the reference
    h = (Kf @ p_cam.t()).t()
    uv = h[:, :2] / (h[:, 2:] + eps)
    out['points_3d'] = p_cam.to(dtype)
    out['normals_3d'] = n_cam.to(dtype)
    out['colors_3d'] = col.to(dtype)
    out['points_2d'] = torch.cat([torch.clamp(uv[:, 0:1], -1, res_x), torch.clamp(uv[:, 1:2], -1, res_y)], dim=-1).to(dtype)
    return out


def project_in_2D(K, camera_pose, points, normals, colors, resolution_px, filter_normals=True, filter_hpr=False, output_nocs=True):
    """projection.py:7-101: camera_pose (4,4) DCM pose; NOCS colours are the object points with x negated (:53-55)."""
    return _project(K, camera_pose, points, normals, colors, resolution_px, filter_normals, filter_hpr, 1 if output_nocs else 0)


def project_in_2D_quat(K, camera_pose, points, normals, colors, resolution_px, filter_normals=False, filter_hpr=False, output_nocs=True):
    """projection.py:104-199: camera_pose = [quaternion(4), translation(3)]; NOCS colours are the object points themselves (:147-149)."""
    dev = points.device
    q, t = camera_pose[:4].to(dev, torch.float32), camera_pose[4:].to(dev, torch.float32)
    bottom = torch.tensor([[0.0, 0.0, 0.0, 1.0]], device=dev)
    pose = torch.cat([torch.cat([qrot_matrix(q), t.view(3, 1)], dim=1), bottom], dim=0)
    return _project(K, pose, points, normals, colors, resolution_px, filter_normals, filter_hpr, 2 if output_nocs else 0)
