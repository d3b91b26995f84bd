"""inside_surfel / inside_circle / inside_circle_opt -- the reference's standalone primitives (sdfrenderer/renderer/primitives.py:4-242) on top
of the HIP splat kernels, same signatures, returning the dense (N[+1], 3, P) weight tensor `prob_color` as the reference does.

`Rasterer.forward` never forms this tensor (the splat kernels composite on the fly; at 256x256 and N = 3 000 it is 0.8 GB per copy); it is
produced here only because a caller of these functions asks for exactly that.  One splat forward pass yields the per-pixel softmax state,
`sdfr_splat_weights` writes the covered entries, and the backward (`sdfr_splat_weights_backward`) is the surfel-centric splat backward fed
with the dense upstream gradient.  Both clamps of every primitive are built (r05): the ones `Rasterer.forward` passes (rasterer.py:92-104:
inside_surfel softclamp=False; inside_circle / inside_circle_opt softclamp=True) and the other ones -- inside_surfel's own default
softclamp=True (the sigmoid mask is positive until exp overflows: practically every surfel covers every pixel, dense work by nature) and the
hard-edged circles (softclamp=False) -- with any positive softclamp_constant.  `grid_2d` must be the renderer's pixel grid (Rasterer.grid:
every pixel of a W x H image, x fastest).
"""
import torch

from .. import _lib


class _WeightsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, vertex_3d, normals, vertex_2d, K, W, H, pid, diam, dconst, add_bg, clamp_alt, clamp_c):
        with _lib.guard(vertex_3d):
            return _WeightsFn._forward(ctx, vertex_3d, normals, vertex_2d, K, W, H, pid, diam, dconst, add_bg, clamp_alt, clamp_c)

    @staticmethod
    def _forward(ctx, vertex_3d, normals, vertex_2d, K, W, H, pid, diam, dconst, add_bg, clamp_alt, clamp_c):
        L = _lib.lib()
        dev = vertex_3d.device
        n = vertex_3d.shape[0]
        f32 = dict(dtype=torch.float32, device=dev)
        P = W * H
        rows = n + (1 if add_bg else 0)
        p_cam = vertex_3d.detach().contiguous().float()
        n_cam = normals.detach().contiguous().float()
        uv = None if pid == 0 else vertex_2d.detach().contiguous().float()
        Kf = K.detach().to(dev, torch.float32).contiguous()
        Kinv = torch.linalg.inv(Kf.cpu()).contiguous().to(dev)                   # primitives.py:204
        eps = torch.finfo(torch.float32).eps
        znorm = bg_logit = None
        bg_argmin = None
        if n > 0:
            z = -p_cam[:, 2]
            if pid != 0:
                znorm = z.norm(p=2).view(1).contiguous()                          # :59 / :142
                zl = torch.clamp(z / (znorm + eps) + 1, min=0) * dconst
            else:
                zl = z * dconst                                                   # :234
            if add_bg:
                bg_logit = (zl.min() - 1).view(1).contiguous()                    # :65 / :147 / :235
                bg_argmin = int(torch.argmin(zl))
        weights = torch.zeros((rows, P), **f32)
        aux = torch.empty((P, 4), **f32)
        if n > 0:
            bbox = torch.empty((n, 4), dtype=torch.int32, device=dev)   # boxes only (no SDFR_PRIM_BINS)
            bg_img = torch.zeros((3, H, W), **f32) if add_bg else None
            st = _lib.stream_ptr()
            _lib.check(L.sdfr_splat_forward_clamp(pid, _lib.ptr(Kf), _lib.ptr(Kinv), _lib.ptr(p_cam), _lib.ptr(n_cam), _lib.ptr(p_cam), _lib.ptr(uv),
                                                  _lib.ptr(znorm), _lib.ptr(bg_img), _lib.ptr(bg_logit), 1, n, None, W, H, diam, dconst, clamp_alt,
                                                  clamp_c, _lib.ptr(bbox), None, None, None, None, _lib.ptr(aux), st), "sdfr_splat_forward_clamp")
            _lib.check(L.sdfr_splat_weights_clamp(pid, _lib.ptr(Kf), _lib.ptr(Kinv), _lib.ptr(p_cam), _lib.ptr(n_cam), _lib.ptr(uv), _lib.ptr(znorm),
                                                  _lib.ptr(bg_logit), 1, n, None, W, H, diam, dconst, clamp_alt, clamp_c, _lib.ptr(aux),
                                                  _lib.ptr(weights), st), "sdfr_splat_weights_clamp")
        elif add_bg:
            weights[0] = 1.0                                                      # only the background row: softmax over one entry
        ctx.save_for_backward(p_cam, n_cam, uv if uv is not None else p_cam, Kf, Kinv, aux, weights,
                              znorm if znorm is not None else p_cam, bg_logit if bg_logit is not None else p_cam)
        ctx.cfg = (n, W, H, pid, diam, dconst, add_bg, bg_argmin, clamp_alt, clamp_c)
        return weights

    @staticmethod
    def backward(ctx, g_w):
        p_cam, n_cam, uv, Kf, Kinv, aux, weights, znorm, bg_logit = ctx.saved_tensors
        n, W, H, pid, diam, dconst, add_bg, bg_argmin, clamp_alt, clamp_c = ctx.cfg
        L = _lib.lib()
        dev = p_cam.device
        f32 = dict(dtype=torch.float32, device=dev)
        g_p, g_n = torch.zeros((max(n, 1), 3), **f32), torch.zeros((max(n, 1), 3), **f32)
        if n > 0:
            g_w = g_w.contiguous().float()
            wsum = (weights * g_w).sum(0).contiguous()                            # softmax backward: sum_j w_j dL/dw_j per pixel
            with _lib.guard(p_cam):
                _lib.check(L.sdfr_splat_weights_backward_clamp(pid, _lib.ptr(Kf), _lib.ptr(Kinv), _lib.ptr(p_cam), _lib.ptr(n_cam),
                                                               _lib.ptr(uv) if pid else None, _lib.ptr(znorm) if pid else None, int(add_bg), 1, n,
                                                               None, W, H, diam, dconst, clamp_alt, clamp_c, _lib.ptr(aux), _lib.ptr(g_w),
                                                               _lib.ptr(wsum), _lib.ptr(g_p), _lib.ptr(g_n), _lib.stream_ptr()),
                           "sdfr_splat_weights_backward_clamp")
            if add_bg and pid != 0:
                # the circle primitives' background logit z.min() - 1 (:65,:147) competes with the surfels' logits (its weight is not 0/1), so the gradient
                # through the min reaches the farthest surfel's depth -- a per-call scalar, as in Rasterer's backward
                eps = torch.finfo(torch.float32).eps
                G = (weights[n] * (g_w[n] - wsum)).sum()
                zq = -p_cam[bg_argmin, 2] / (znorm[0] + eps) + 1
                g_p[bg_argmin, 2] += torch.where(zq >= 0, -G * dconst / (znorm[0] + eps), torch.zeros_like(G))
        return g_p[:n], g_n[:n], None, None, None, None, None, None, None, None, None, None


def _resolution_from_grid(grid_2d):
    g = grid_2d.reshape(-1, 2)
    W, H = int(g[:, 0].max()) + 1, int(g[:, 1].max()) + 1
    ok = g.shape[0] == W * H
    if ok:
        idx = torch.arange(W * H, device=g.device)
        ok = bool(torch.equal(g[:, 0].long(), idx % W) and torch.equal(g[:, 1].long(), idx // W))
    if not ok:
        raise NotImplementedError("grid_2d must be the renderer's pixel grid (every pixel of a W x H image, x fastest: Rasterer.grid)")
    return W, H


def _finish(weights, dtype):
    return weights.to(dtype).unsqueeze(1).expand(-1, 3, -1)                      # primitives.py:71 / :162 / :241-242


def _clamp_constant(softclamp_constant):
    c = float(softclamp_constant)
    if not (c > 0.0):
        # sigmoid((r - d) * c) > 0 with c <= 0 covers everything BEYOND r + 88.7 / |c| (or every pixel for c = 0): not a primitive anyone renders
        raise NotImplementedError("softclamp_constant must be positive")
    return c


def inside_surfel(K, grid_2d, vertex_2d, vertex_3d, normals, diam=0.03, depth_constant=150, softclamp=True, softclamp_constant=5,
                  add_bg=True):
    """primitives.py:165-242: tangent discs.  softclamp=False: the hard disc edge Rasterer.forward passes (:220); softclamp=True (the
    function's own default): mask = sigmoid((diam - d) * softclamp_constant) > 0 (:217-218,:226)."""
    _lib.require_gpu_float(vertex_3d, normals)
    W, H = _resolution_from_grid(grid_2d)
    return _finish(_WeightsFn.apply(vertex_3d, normals, None, K, W, H, 0, float(diam), float(depth_constant), bool(add_bg),
                                    1 if softclamp else 0, _clamp_constant(softclamp_constant) if softclamp else 5.0), K.dtype)


def inside_circle(K, grid_2d, vertex_2d, vertex_3d, normals, diam=0.07, depth_constant=100, softclamp=True, softclamp_constant=3,
                  add_bg=False):
    """primitives.py:4-71: 2-D circles.  softclamp=True: coverage = sigmoid((r - d) * softclamp_constant) > 0, i.e. out to where exp
    overflows (:46-49); softclamp=False: clamp(r - d, min=0) > 0, the hard circle (:51-53)."""
    _lib.require_gpu_float(vertex_3d, vertex_2d)
    W, H = _resolution_from_grid(grid_2d)
    return _finish(_WeightsFn.apply(vertex_3d, normals, vertex_2d, K, W, H, 1, float(diam), float(depth_constant), bool(add_bg),
                                    0 if softclamp else 1, _clamp_constant(softclamp_constant) if softclamp else 3.0), K.dtype)


def inside_circle_opt(K, grid_2d, vertex_2d, vertex_3d, normals, diam=0.06, depth_constant=10000, softclamp=True, softclamp_constant=5,
                      add_bg=True):
    """primitives.py:74-162: every vertex stamps the 15x15 offsets `grid_2d` (Rasterer.grid_prim); image size from K (:109-110).
    softclamp=True: every stamped pixel is covered (the sigmoid of :118 is positive on the whole stamp for any positive constant that keeps
    exp finite there); softclamp=False: only the offsets inside the circle of radius K00 diam / z count (:120)."""
    if grid_2d.reshape(-1, 2).shape[0] != 225:
        raise NotImplementedError("grid_2d must be the 15x15 stamp offsets (Rasterer.grid_prim)")
    _lib.require_gpu_float(vertex_3d, vertex_2d)
    W, H = int(K[0, 2]) * 2, int(K[1, 2]) * 2
    c = _clamp_constant(softclamp_constant) if softclamp else 5.0
    if softclamp and c * 9.9 >= 88.0:
        # (r - |offset|) * c > -88.7 must hold on the whole 15x15 stamp (|offset| <= 9.9) for "every stamped pixel is covered"
        raise NotImplementedError("inside_circle_opt: softclamp_constant >= 8.9 lets the sigmoid underflow inside the stamp")
    return _finish(_WeightsFn.apply(vertex_3d, normals, vertex_2d, K, W, H, 2, float(diam), float(depth_constant), bool(add_bg),
                                    0 if softclamp else 1, c), K.dtype)
