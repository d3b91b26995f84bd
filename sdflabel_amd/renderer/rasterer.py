"""Rasterer -- nn.Module facade of the HIP surfel renderer (mirror of the reference sdfrenderer/renderer/rasterer.py:9-155).

Same constructor and forward signature; returns (rendering: dict, points: dict) or `rendering`.  Projection
(sdflabel_amd/csrc/project.hip) and splat/composite (sdflabel_amd/csrc/splat.hip) run behind ONE autograd.Function, so the
backward is two kernel launches (surfel-centric splat backward, then projection backward with the pose reduction).
"""
import numpy as np
import torch

from .. import _lib
from .utils_rasterer import calibration_matrix, qrot_matrix

_DIAM_DISC = 0.04            # rasterer.py:102-104
_DEPTH_CONSTANT = 150.0      # primitives.py:171


class _RasterFn(torch.autograd.Function):
    """(coords, normals, colors, pose44) -> color, mask, depth, normals_img, p_cam, col  (+ fidx as a plain attribute)."""

    @staticmethod
    def forward(ctx, coords, normals, colors, pose, K, Kinv, res, nocs_mode, want_mask, want_depth, want_normals, want_filter, holder):
        L = _lib.lib()
        W, H = res
        dev = coords.device
        n = coords.shape[0]
        f32 = dict(dtype=torch.float32, device=dev)
        coords_c = coords.detach().contiguous()
        normals_c = normals.detach().contiguous()
        colors_c = None if nocs_mode else colors.detach().contiguous()
        pose_c = pose.detach().contiguous().float()
        m = max(n, 1)
        p_cam = torch.empty((m, 3), **f32)
        n_cam = torch.empty((m, 3), **f32)
        col = torch.empty((m, 3), **f32)
        uv = torch.empty((m, 2), **f32)
        fidx = torch.empty((m,), dtype=torch.int32, device=dev) if want_filter else None
        fcnt = torch.zeros((1,), dtype=torch.int32, device=dev) if want_filter else None
        st = _lib.stream_ptr()
        if n > 0:
            _lib.check(L.sdfr_project_dcm(_lib.ptr(pose_c), _lib.ptr(K), _lib.ptr(coords_c), _lib.ptr(normals_c), _lib.ptr(colors_c), 1, n,
                                          None, int(nocs_mode), W, H, _lib.ptr(p_cam), _lib.ptr(n_cam), _lib.ptr(col), _lib.ptr(uv),
                                          _lib.ptr(fidx), _lib.ptr(fcnt), st), "sdfr_project_dcm")
        attr = ((col + 1) / 2) if nocs_mode else col                       # rasterer.py:113-116
        attr = attr.contiguous()
        color = torch.empty((3, H, W), **f32)
        mask = torch.empty((1, H, W), **f32) if want_mask else None
        depth = torch.empty((1, H, W), **f32) if want_depth else None
        nimg = torch.empty((3, H, W), **f32) if want_normals else None
        aux = torch.empty((H * W, 4), **f32)
        bbox = torch.empty((m, 4), dtype=torch.int32, device=dev)
        _lib.check(L.sdfr_splat_forward(_lib.ptr(K), _lib.ptr(Kinv), _lib.ptr(p_cam), _lib.ptr(n_cam), _lib.ptr(attr), 1, n, None, W, H,
                                        _DIAM_DISC, _DEPTH_CONSTANT, _lib.ptr(bbox), _lib.ptr(color), _lib.ptr(mask), _lib.ptr(depth),
                                        _lib.ptr(nimg), _lib.ptr(aux), st), "sdfr_splat_forward")
        nf = int(fcnt.item()) if (want_filter and n > 0) else 0
        holder["fidx"] = fidx[:nf].long() if want_filter else None
        holder["uv"] = uv[:n]
        ctx.save_for_backward(coords_c, normals_c, pose_c, K, Kinv, p_cam, n_cam, attr, aux, color,
                              mask if want_mask else color, depth if want_depth else color, nimg if want_normals else color)
        ctx.cfg = (n, W, H, nocs_mode, want_mask, want_depth, want_normals)
        outs = (color, mask if want_mask else color.new_zeros(()), depth if want_depth else color.new_zeros(()),
                nimg if want_normals else color.new_zeros(()), p_cam[:n], n_cam[:n], col[:n])
        ctx.mark_non_differentiable(outs[5])
        return outs

    @staticmethod
    def backward(ctx, g_color, g_mask, g_depth, g_nimg, g_pcam_ext, _g_ncam, g_col_ext):
        L = _lib.lib()
        coords, normals, pose, K, Kinv, p_cam, n_cam, attr, aux, color, mask, depth, nimg = ctx.saved_tensors
        n, W, H, nocs_mode, want_mask, want_depth, want_normals = ctx.cfg
        dev = coords.device
        f32 = dict(dtype=torch.float32, device=dev)
        m = max(n, 1)
        st = _lib.stream_ptr()

        def cg(g, want):
            return g.contiguous().float() if (want and g is not None) else None

        g_color = cg(g_color, True)
        g_mask = cg(g_mask, want_mask)
        g_depth = cg(g_depth, want_depth)
        g_nimg = cg(g_nimg, want_normals)
        g_p = torch.zeros((m, 3), **f32)
        g_n = torch.zeros((m, 3), **f32)
        g_a = torch.zeros((m, 3), **f32)
        if n > 0:
            _lib.check(L.sdfr_splat_backward(_lib.ptr(K), _lib.ptr(Kinv), _lib.ptr(p_cam), _lib.ptr(n_cam), _lib.ptr(attr), 1, n, None, W, H,
                                             _DIAM_DISC, _DEPTH_CONSTANT, _lib.ptr(aux), _lib.ptr(color), _lib.ptr(mask), _lib.ptr(depth),
                                             _lib.ptr(nimg), _lib.ptr(g_color), _lib.ptr(g_mask), _lib.ptr(g_depth), _lib.ptr(g_nimg),
                                             _lib.ptr(g_p), _lib.ptr(g_n), _lib.ptr(g_a), st), "sdfr_splat_backward")
        g_col = g_a * 0.5 if nocs_mode else g_a                              # attr = (col+1)/2
        if g_col_ext is not None:
            g_col = g_col[:n] + g_col_ext
        if g_pcam_ext is not None:
            g_p = g_p[:n] + g_pcam_ext
        g_p = g_p.contiguous()
        g_col = g_col.contiguous()
        g_points = torch.zeros((m, 3), **f32)
        g_normals = torch.zeros((m, 3), **f32)
        g_colors = None if nocs_mode else torch.zeros((m, 3), **f32)
        g_pose = torch.zeros((4, 4), **f32)
        if n > 0:
            _lib.check(L.sdfr_project_dcm_bwd(_lib.ptr(pose), _lib.ptr(coords), _lib.ptr(normals), _lib.ptr(g_p), _lib.ptr(g_n),
                                              _lib.ptr(g_col), 1, n, None, int(nocs_mode), _lib.ptr(g_points), _lib.ptr(g_normals),
                                              _lib.ptr(g_colors), _lib.ptr(g_pose), st), "sdfr_project_dcm_bwd")
        return (g_points[:n], g_normals[:n], None if nocs_mode else g_colors[:n], g_pose) + (None,) * 9


class Rasterer(torch.nn.Module):
    def __init__(self, K, resolution_px, diagonal_mm=20, focal_len_mm=70, precision=torch.float32):
        """K (3,3) intrinsics or None (then derived from sensor diagonal / focal length); resolution_px = (W, H)."""
        super().__init__()
        self.res_x_px, self.res_y_px = resolution_px
        yy, xx = np.mgrid[0:self.res_y_px, 0:self.res_x_px]
        self.register_buffer('grid', torch.from_numpy(np.stack((xx, yy), axis=-1).reshape((1, -1, 2))))
        if K is None:
            K = torch.from_numpy(calibration_matrix((self.res_x_px, self.res_y_px), diagonal_mm, focal_len_mm, skew=0))
        if precision != torch.float32:
            raise NotImplementedError("sdflabel_amd renders in float32 (requested %s)" % precision)
        K = K.detach().to(torch.float32)
        self.register_buffer('K', K.contiguous())
        # K^-1 in float32 exactly as the reference computes it on every call (primitives.py:204), once, on the host
        self.register_buffer('Kinv', torch.linalg.inv(K.cpu().float()).contiguous())

    def forward(self, coords, normals, colors, camera_matrix, rot='quat', primitives='disc', bg=None, output_mask=False,
                output_depth=False, output_normals=False, output_nocs=False, output_points=True):
        _lib.require_gpu_f32(coords, normals, None if output_nocs else colors)
        if primitives != 'disc':
            raise NotImplementedError("primitives='%s': only the 3-D tangent disc ('disc', the optimizer's primitive) is built" % primitives)
        if bg is not None:
            raise NotImplementedError("background compositing (bg=...) is not built yet")
        dev = coords.device
        K = self.K.to(dev)
        Kinv = self.Kinv.to(dev)
        if rot == 'dcm':
            pose = camera_matrix.to(dev, torch.float32)
            nocs_mode = 1 if output_nocs else 0                 # NOCS colour = p * (-1,1,1), projection.py:53-55
            want_filter = True
        elif rot == 'quat':
            q, t = camera_matrix[:4].to(dev, torch.float32), camera_matrix[4:].to(dev, torch.float32)
            pose = torch.eye(4, dtype=torch.float32, device=dev)
            pose = torch.cat([torch.cat([qrot_matrix(q), t.view(3, 1)], dim=1), pose[3:]], dim=0)
            nocs_mode = 2 if output_nocs else 0                 # x not flipped, projection.py:147-149
            want_filter = False                                 # filter_normals=False default, projection.py:105
        else:
            raise ValueError("rot must be 'dcm' or 'quat'")
        if coords.shape[0] != normals.shape[0]:
            raise _lib.SdfrError("coords and normals must have the same number of rows")
        holder = {}
        color, mask, depth, nimg, p_cam, n_cam, col = _RasterFn.apply(
            coords, normals, colors if not output_nocs else None, pose, K, Kinv, (self.res_x_px, self.res_y_px), nocs_mode,
            bool(output_mask), bool(output_depth), bool(output_normals), want_filter, holder)
        rendering = {'color': color}
        if output_mask:
            rendering['mask'] = mask
        if output_depth:
            rendering['depth'] = depth
        if output_normals:
            rendering['normals'] = nimg
        if output_points:
            if not want_filter:
                raise KeyError('points_3d_filt')            # same failure as the reference for rot='quat' (rasterer.py:151)
            fidx = holder["fidx"]
            points = {'xyz': p_cam, 'rgb': (col + 1) / 2,
                      'xyzf': p_cam.index_select(0, fidx), 'rgbf': (col.index_select(0, fidx) + 1) / 2}
            return rendering, points
        return rendering
