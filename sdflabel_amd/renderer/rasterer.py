"""Rasterer -- nn.Module facade of the HIP surfel renderer (mirror of the reference sdfrenderer/renderer/rasterer.py:9-155).

Same constructor and forward signature; returns (rendering: dict, points: dict) or `rendering`.  Projection
(sdflabel_amd/csrc/project.hip) and splat/composite (sdflabel_amd/csrc/splat.hip) run behind ONE autograd.Function, so the
backward is two kernel launches (surfel-centric splat backward, then projection backward with the pose reduction).
"""
import numpy as np
import torch

from .. import _lib
from .utils_rasterer import calibration_matrix, qrot_matrix

# (primitive id, diam, depth_constant): the constants Rasterer.forward passes / leaves at their defaults (rasterer.py:92-104)
_PRIMS = {'disc': (0, 0.04, 150.0), 'circle': (1, 0.02, 100.0), 'circle_opt': (2, 0.025, 10000.0)}


class _RasterFn(torch.autograd.Function):
    """(coords, normals, colors, pose44, bg) -> color, mask, depth, normals_img, p_cam, n_cam, col  (+ fidx through `holder`)."""

    @staticmethod
    @_lib.traced("Rasterer.forward")
    def forward(ctx, coords, *args):
        with _lib.guard(coords):                      # launch stream / allocations of the device that holds the surfels
            return _RasterFn._forward(ctx, coords, *args)

    @staticmethod
    @_lib.traced("Rasterer.backward")
    def backward(ctx, *grads):
        with _lib.guard(ctx.saved_tensors[0]):
            return _RasterFn._backward(ctx, *grads)

    @staticmethod
    def _forward(ctx, coords, normals, colors, pose, bg, K, Kinv, res, nocs_mode, prim, half_attr, want_mask, want_depth, want_normals,
                 want_filter, holder):
        L = _lib.lib()
        W, H = res
        pid, diam, dconst = _PRIMS[prim]
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
                                          _lib.ptr(fidx), _lib.ptr(fcnt), None, None, st), "sdfr_project_dcm")
        attr = ((col + 1) / 2) if half_attr else col                       # rasterer.py:108-109,113-116
        attr = attr.contiguous()
        # per-crop scalars of the secondary primitives / the background row (tiny host-layer reductions)
        znorm = bg_logit = bg_c = None
        eps = torch.finfo(torch.float32).eps
        if n > 0 and (pid != 0 or bg is not None):
            z = -p_cam[:n, 2]
            if pid != 0:
                znorm = z.norm(p=2).view(1).contiguous()                   # primitives.py:59 / :142 (detached)
                zl = torch.clamp(z / (znorm + eps) + 1, min=0) * dconst
            else:
                zl = z * dconst                                            # primitives.py:234
            if bg is not None:
                bg_logit = (zl.min() - 1).view(1).contiguous()             # primitives.py:65 / :147 / :235
                holder["bg_argmin"] = int(torch.argmin(zl))
        elif pid != 0:
            znorm = torch.ones((1,), **f32)
        if bg is not None:
            bg_c = bg.detach().to(dev, torch.float32).reshape(3, H, W).contiguous()
            if bg_logit is None:
                bg_logit = torch.zeros((1,), **f32)
        color = torch.empty((3, H, W), **f32)
        mask = torch.empty((1, H, W), **f32) if want_mask else None
        depth = torch.empty((1, H, W), **f32) if want_depth else None
        nimg = torch.empty((3, H, W), **f32) if want_normals else None
        aux = torch.empty((H * W, 4), **f32)
        bbox = torch.empty((m, 4), dtype=torch.int32, device=dev)       # boxes only (no SDFR_PRIM_BINS: one crop per call)
        _lib.check(L.sdfr_splat_forward(pid, _lib.ptr(K), _lib.ptr(Kinv), _lib.ptr(p_cam), _lib.ptr(n_cam), _lib.ptr(attr), _lib.ptr(uv),
                                        _lib.ptr(znorm), _lib.ptr(bg_c), _lib.ptr(bg_logit), 1, n, None, W, H, diam, dconst,
                                        _lib.ptr(bbox), _lib.ptr(color), _lib.ptr(mask), _lib.ptr(depth), _lib.ptr(nimg), _lib.ptr(aux), st),
                   "sdfr_splat_forward")
        nf = int(fcnt.item()) if (want_filter and n > 0) else 0
        holder["fidx"] = fidx[:nf].long() if want_filter else None
        dummy = color
        ctx.save_for_backward(coords_c, normals_c, pose_c, K, Kinv, p_cam, n_cam, attr, aux, color,
                              mask if want_mask else dummy, depth if want_depth else dummy, nimg if want_normals else dummy, uv,
                              znorm if znorm is not None else dummy, bg_c if bg_c is not None else dummy,
                              bg_logit if bg_logit is not None else dummy)
        ctx.cfg = (n, W, H, nocs_mode, prim, half_attr, want_mask, want_depth, want_normals, bg is not None, holder.get("bg_argmin"))
        outs = (color, mask if want_mask else color.new_zeros(()), depth if want_depth else color.new_zeros(()),
                nimg if want_normals else color.new_zeros(()), p_cam[:n], n_cam[:n], col[:n])
        ctx.mark_non_differentiable(outs[5])
        return outs

    @staticmethod
    def _backward(ctx, g_color, g_mask, g_depth, g_nimg, g_pcam_ext, _g_ncam, g_col_ext):
        L = _lib.lib()
        (coords, normals, pose, K, Kinv, p_cam, n_cam, attr, aux, color, mask, depth, nimg, uv, znorm, bg_c, bg_logit) = ctx.saved_tensors
        n, W, H, nocs_mode, prim, half_attr, want_mask, want_depth, want_normals, has_bg, bg_argmin = ctx.cfg
        pid, diam, dconst = _PRIMS[prim]
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
            _lib.check(L.sdfr_splat_backward(pid, _lib.ptr(K), _lib.ptr(Kinv), _lib.ptr(p_cam), _lib.ptr(n_cam), _lib.ptr(attr),
                                             _lib.ptr(uv), _lib.ptr(znorm) if pid else None, _lib.ptr(bg_c) if has_bg else None,
                                             _lib.ptr(bg_logit) if has_bg else None, 1, n, None, W, H, diam, dconst, _lib.ptr(aux),
                                             _lib.ptr(color), _lib.ptr(mask), _lib.ptr(depth), _lib.ptr(nimg), _lib.ptr(g_color),
                                             _lib.ptr(g_mask), _lib.ptr(g_depth), _lib.ptr(g_nimg), _lib.ptr(g_p), _lib.ptr(g_n),
                                             _lib.ptr(g_a), st), "sdfr_splat_backward")
            if has_bg and pid != 0:
                # the circle primitives' background logit z.min() - 1 (primitives.py:65,147) competes with the surfels' logits (it sits one
                # unit below the farthest surfel's), so its weight is not 0/1 and the gradient through the min reaches the farthest surfel's
                # depth; a per-crop scalar, done here.  (disc: the background sits ~500 below every surfel logit -- weight exactly 0 or 1.)
                eps = torch.finfo(torch.float32).eps
                gates = aux[:, 3].contiguous().view(torch.int32)
                w_bg = torch.exp(bg_logit - aux[:, 1]) / aux[:, 2]
                gc = g_color.view(3, -1) * torch.stack([(gates & 1) != 0, (gates & 2) != 0, (gates & 4) != 0]).float()
                dLdw = (gc * bg_c.view(3, -1)).sum(0)
                S = (gc * color.view(3, -1)).sum(0)
                if g_mask is not None:
                    gm = g_mask.view(-1) * ((gates & 8) != 0).float()
                    dLdw = dLdw + gm
                    S = S + gm * mask.view(-1)
                if g_depth is not None:
                    S = S + g_depth.view(-1) * depth.view(-1)
                if g_nimg is not None:
                    gn = g_nimg.view(3, -1) * torch.stack([(gates & 16) != 0, (gates & 32) != 0, (gates & 64) != 0]).float()
                    S = S + (gn * nimg.view(3, -1)).sum(0)
                G = (w_bg * (dLdw - S)).sum()
                zq = -p_cam[bg_argmin, 2] / (znorm[0] + eps) + 1
                g_p[bg_argmin, 2] += torch.where(zq >= 0, -G * dconst / (znorm[0] + eps), torch.zeros_like(G))
        g_col = g_a * 0.5 if half_attr else g_a                              # attr = (col + 1) / 2
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
                                              _lib.ptr(g_colors), _lib.ptr(g_pose), None, None, st), "sdfr_project_dcm_bwd")
        return (g_points[:n], g_normals[:n], None if nocs_mode else g_colors[:n], g_pose, None) + (None,) * 11


_PLACEHOLDERS = {}


def _placeholder(dev, slot):
    """the 0-d zero tensor standing in for an image that was not asked for: one per (device, output slot), made once (r05: a fresh
    new_zeros(()) per call was a fill launch per iteration); non-differentiable and never written, so sharing it across calls is safe, and
    distinct per slot"""
    key = (dev.type, dev.index, slot)
    t = _PLACEHOLDERS.get(key)
    if t is None:
        t = _PLACEHOLDERS[key] = torch.zeros((), dtype=torch.float32, device=dev)
    return t


class _RasterDiscFn(torch.autograd.Function):
    """The optimizer's configuration (pipelines/optimizer.py:110-123: rot='dcm', primitives='disc', bg=None, output_nocs=True) with the fused
    kernels of the batched path: ONE projection launch writes the camera-frame surfels, the composited attribute (c + 1) / 2, the front-facing
    rows (points['xyzf']) and the surfel -> front-slot map; the backward is the splat backward + ONE projection backward that folds in the
    1/2 of the attribute map and the gradient arriving through xyzf.  5 launches forward, 2 backward, one host synchronisation (the number
    of front-facing surfels is a tensor SHAPE of the reference's API).  Same kernels, same bits as _RasterFn.

    (coords, normals, pose44) -> color, mask, depth, normals_img, p_cam, n_cam, rgb (= attr), xyzf, rgbf"""

    @staticmethod
    def forward(ctx, coords, normals, pose, K, Kinv, res, nocs_mode, want_mask, want_depth, want_normals):
        ctx.set_materialize_grads(False)
        L = _lib.lib()
        W, H = res
        dev = coords.device
        n = coords.shape[0]
        m = max(n, 1)
        f32 = dict(dtype=torch.float32, device=dev)
        with _lib.guard(coords):
            coords_c, normals_c = coords.detach().contiguous(), normals.detach().contiguous()
            pose_c = pose.detach().contiguous().float()
            slab = torch.empty((5, m, 3), **f32)                      # p_cam, n_cam, attr, xyzf, rgbf
            p_cam, n_cam, attr, xyzf, rgbf = slab[0], slab[1], slab[2], slab[3], slab[4]
            # fidx | fslot | fcnt -- not zeroed (r05): with n > 0 the projection kernel writes fcnt and every fslot[s < n]; with n = 0 none is read
            ints = torch.empty((2 * m + 1,), dtype=torch.int32, device=dev)
            fidx, fslot, fcnt = ints[:m], ints[m:2 * m], ints[2 * m:]
            imgs = torch.empty((8, H, W), **f32)                      # color(3) | mask | depth | normals(3)
            color, mask, depth, nimg = imgs[0:3], imgs[3:4], imgs[4:5], imgs[5:8]
            aux = torch.empty((H * W, 4), **f32)
            bbox = torch.empty((m, 4), dtype=torch.int32, device=dev)
            st = _lib.stream_ptr()
            nf = 0
            if n > 0:
                _lib.check(L.sdfr_project_dcm(_lib.ptr(pose_c), _lib.ptr(K), _lib.ptr(coords_c), _lib.ptr(normals_c), None, 1, n, None,
                                              int(nocs_mode) | 4, W, H, _lib.ptr(p_cam), _lib.ptr(n_cam), _lib.ptr(attr), None, _lib.ptr(fidx),
                                              _lib.ptr(fcnt), _lib.ptr(xyzf), _lib.ptr(fslot), st), "sdfr_project_dcm")
                _lib.check(L.sdfr_gather_rows3(_lib.ptr(rgbf), _lib.ptr(attr), _lib.ptr(fidx), 1, n, _lib.ptr(fcnt), st), "sdfr_gather_rows3")
            _lib.check(L.sdfr_splat_forward(0, _lib.ptr(K), _lib.ptr(Kinv), _lib.ptr(p_cam), _lib.ptr(n_cam), _lib.ptr(attr), None, None, None, None,
                                            1, n, None, W, H, _PRIMS['disc'][1], _PRIMS['disc'][2], _lib.ptr(bbox), _lib.ptr(color),
                                            _lib.ptr(mask) if want_mask else None, _lib.ptr(depth) if want_depth else None,
                                            _lib.ptr(nimg) if want_normals else None, _lib.ptr(aux), st), "sdfr_splat_forward")
            if n > 0:
                nf = int(fcnt.item())                                 # the one synchronisation: xyzf / rgbf are (N_f, 3) tensors
        ctx.save_for_backward(coords_c, normals_c, pose_c, K, Kinv, slab, imgs, aux, fslot)
        ctx.cfg = (n, nf, W, H, int(nocs_mode), want_mask, want_depth, want_normals)
        # (images that were not asked for: DISTINCT placeholder tensors, marked non-differentiable -- the same object in several output slots
        #  would make autograd route a gradient for any of them to the last duplicate)
        zm, zd, zn = (None if want_mask else _placeholder(dev, 0)), (None if want_depth else _placeholder(dev, 1)), (None if want_normals else _placeholder(dev, 2))
        outs = (color, mask if want_mask else zm, depth if want_depth else zd, nimg if want_normals else zn, p_cam[:n], n_cam[:n], attr[:n],
                xyzf[:nf], rgbf[:nf])
        ctx.mark_non_differentiable(outs[5], *[z for z in (zm, zd, zn) if z is not None])
        return outs

    @staticmethod
    @_lib.traced("Rasterer.backward")
    def backward(ctx, g_color, g_mask, g_depth, g_nimg, g_pcam_ext, _g_ncam, g_rgb, g_xyzf, g_rgbf):
        L = _lib.lib()
        coords, normals, pose, K, Kinv, slab, imgs, aux, fslot = ctx.saved_tensors
        n, nf, W, H, nocs_mode, want_mask, want_depth, want_normals = ctx.cfg
        dev = coords.device
        f32 = dict(dtype=torch.float32, device=dev)
        m = max(n, 1)
        p_cam, n_cam, attr = slab[0], slab[1], slab[2]
        color, mask, depth, nimg = imgs[0:3], imgs[3:4], imgs[4:5], imgs[5:8]

        def cg(g, want):
            return g.contiguous().float() if (want and g is not None) else None

        with _lib.guard(coords):
            g_color, g_mask, g_depth, g_nimg = cg(g_color, True), cg(g_mask, want_mask), cg(g_depth, want_depth), cg(g_nimg, want_normals)
            gs = torch.empty((5, m, 3), **f32)                        # g_p_cam, g_n_cam, g_attr | g_points, g_normals
            g_p, g_n, g_a, g_points, g_normals = gs[0], gs[1], gs[2], gs[3], gs[4]
            g_pose = torch.empty((4, 4), **f32)
            st = _lib.stream_ptr()
            if n == 0:
                return (gs[3][:0], gs[4][:0], torch.zeros((4, 4), **f32)) + (None,) * 7
            if g_color is None and g_mask is None and g_depth is None and g_nimg is None:
                gs[:3].zero_()
            else:
                _lib.check(L.sdfr_splat_backward(0, _lib.ptr(K), _lib.ptr(Kinv), _lib.ptr(p_cam), _lib.ptr(n_cam), _lib.ptr(attr), None, None, None,
                                                 None, 1, n, None, W, H, _PRIMS['disc'][1], _PRIMS['disc'][2], _lib.ptr(aux), _lib.ptr(color),
                                                 _lib.ptr(mask), _lib.ptr(depth), _lib.ptr(nimg), _lib.ptr(g_color), _lib.ptr(g_mask),
                                                 _lib.ptr(g_depth), _lib.ptr(g_nimg), _lib.ptr(g_p), _lib.ptr(g_n), _lib.ptr(g_a), st),
                           "sdfr_splat_backward")
            # gradients arriving through the point outputs other than xyzf (not used by the optimizer's loop: plain torch ops)
            if g_pcam_ext is not None:
                g_p[:n] += g_pcam_ext
            if g_rgb is not None:
                g_a[:n] += g_rgb
            if g_rgbf is not None and nf > 0:
                fs = fslot[:n].long()
                g_a[:n] += torch.where((fs >= 0).unsqueeze(1), g_rgbf.contiguous().float()[fs.clamp(min=0)], torch.zeros((), **f32))
            g_xyzf = None if (g_xyzf is None or nf == 0) else g_xyzf.contiguous().float()
            _lib.check(L.sdfr_project_dcm_bwd(_lib.ptr(pose), _lib.ptr(coords), _lib.ptr(normals), _lib.ptr(g_p), _lib.ptr(g_n), _lib.ptr(g_a), 1, n,
                                              None, nocs_mode | 4, _lib.ptr(g_points), _lib.ptr(g_normals), None, _lib.ptr(g_pose), _lib.ptr(g_xyzf),
                                              _lib.ptr(fslot), st), "sdfr_project_dcm_bwd")
        return (g_points[:n], g_normals[:n], g_pose) + (None,) * 7


class Rasterer(torch.nn.Module):
    def __init__(self, K, resolution_px, diagonal_mm=20, focal_len_mm=70, precision=torch.float32):
        """K (3,3) intrinsics or None (then derived from sensor diagonal / focal length); resolution_px = (W, H)."""
        super().__init__()
        self.res_x_px, self.res_y_px = resolution_px
        yy, xx = np.mgrid[0:self.res_y_px, 0:self.res_x_px]
        self.register_buffer('grid', torch.from_numpy(np.stack((xx, yy), axis=-1).reshape((1, -1, 2))))
        # the 15x15 stamp offsets of inside_circle_opt (rasterer.py:29-32): the kernels generate them on the fly, the buffer exists so that
        # state_dict() has the reference's keys (grid, grid_prim, K)
        yy, xx = np.mgrid[-7:8, -7:8]
        self.register_buffer('grid_prim', torch.from_numpy(np.stack((xx, yy), axis=-1).reshape((1, -1, 2))))
        if K is None:
            K = torch.from_numpy(calibration_matrix((self.res_x_px, self.res_y_px), diagonal_mm, focal_len_mm, skew=0))
        if precision not in (torch.float32, torch.float16):
            raise NotImplementedError("sdflabel_amd renders for float32 or float16 callers (requested %s)" % precision)
        # the kernels always compute in float32; `precision` only says what the caller's tensors are (half inputs are widened at the
        # boundary, results narrowed back to the input dtype)
        K = K.detach().to(torch.float32)
        self.register_buffer('K', K.contiguous())
        # K^-1 in float32 exactly as the reference computes it on every call (primitives.py:204), once, on the host
        # (not part of the state_dict: the reference has no such buffer)
        self.register_buffer('Kinv', torch.linalg.inv(K.cpu().float()).contiguous(), persistent=False)
        self.fast_path = True           # False: always the general autograd.Function (tests compare the two)

    @_lib.traced("Rasterer.forward")
    def forward(self, coords, normals, colors, camera_matrix, rot='quat', primitives='disc', bg=None, output_mask=False,
                output_depth=False, output_normals=False, output_nocs=False, output_points=True):
        _lib.require_gpu_float(coords, normals, None if output_nocs else colors)
        out_dtype = coords.dtype
        if out_dtype != torch.float32:
            coords, normals = coords.float(), normals.float()
            colors = None if (colors is None or output_nocs) else colors.float()
        if primitives not in _PRIMS:
            raise ValueError("primitives must be 'disc', 'circle' or 'circle_opt'")
        if bg is not None and (output_depth or output_normals):
            # the reference fails here as well: its depth / normals products mix N+1 weight rows with N attribute rows
            # (rasterer.py:108,134-143)
            raise RuntimeError("bg is incompatible with output_depth / output_normals (shape mismatch in the reference, rasterer.py:108-143)")
        if primitives == 'circle_opt' and (int(self.K[0, 2]) * 2 != self.res_x_px or int(self.K[1, 2]) * 2 != self.res_y_px):
            raise RuntimeError("circle_opt derives the image size from K's principal point (primitives.py:109-110); it must equal the resolution")
        dev = coords.device
        K = self.K if self.K.device == dev else self.K.to(dev)
        Kinv = self.Kinv if self.Kinv.device == dev else self.Kinv.to(dev)
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
        if rot == 'dcm' and primitives == 'disc' and bg is None and output_nocs and self.fast_path:
            # the optimizer's configuration: fused projection / fewer launches, same kernels and bits as the general path below
            color, mask, depth, nimg, p_cam, n_cam, rgb, xyzf, rgbf = _RasterDiscFn.apply(
                coords, normals, pose, K, Kinv, (self.res_x_px, self.res_y_px), nocs_mode, bool(output_mask), bool(output_depth),
                bool(output_normals))
            if out_dtype != torch.float32:
                color, mask, depth, nimg, p_cam, rgb, xyzf, rgbf = (t.to(out_dtype) for t in (color, mask, depth, nimg, p_cam, rgb, xyzf, rgbf))
            rendering = {'color': color}
            if output_mask:
                rendering['mask'] = mask
            if output_depth:
                rendering['depth'] = depth
            if output_normals:
                rendering['normals'] = nimg
            if output_points:
                return rendering, {'xyz': p_cam, 'rgb': rgb, 'xyzf': xyzf, 'rgbf': rgbf}
            return rendering
        holder = {}
        half_attr = bool(output_nocs) or (bg is not None)       # (c+1)/2 for NOCS (:114) and always with a background (:109)
        color, mask, depth, nimg, p_cam, n_cam, col = _RasterFn.apply(
            coords, normals, colors if not output_nocs else None, pose, bg, K, Kinv, (self.res_x_px, self.res_y_px), nocs_mode,
            primitives, half_attr, bool(output_mask), bool(output_depth), bool(output_normals), want_filter, holder)
        if out_dtype != torch.float32:
            color, mask, depth, nimg, p_cam, col = (t.to(out_dtype) for t in (color, mask, depth, nimg, p_cam, col))
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
