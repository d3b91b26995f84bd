"""SphereTracer -- per-ray sphere tracing of the DeepSDF level set (render mode of BASELINE.json's north_star wording).

NOT part of the reference: TRI-ML/sdflabel renders by splatting the surfels of a grid band (sdflabel_amd.Rasterer reproduces that to 1e-4).
This class is the other classic way to render an SDF, built on the same decoder kernels, offered beside the faithful path and labelled as a
different algorithm: there is no reference output it could be checked against, so its tests are self-consistency (|sdf| at the hits, agreement
with the splat renderer's silhouette / depth / NOCS up to the band thickness, gradients against finite differences).

    forward(yaw[B], trans[B,3], latent[B,L]) -> {'color' (NOCS) [B,3,H,W], 'mask' [B,1,H,W], 'depth' [B,1,H,W], 'normals' [B,3,H,W]}

March (csrc/trace.hip): every pixel's ray is clipped against the object cube, then `steps` times: decoder on the ACTIVE rays only
(sdfr_mlp_forward_counted reads the count on the device) -> advance by the decoder value -> retire hits (|sdf| < eps) and exits, compact the
rest with wave ballots.  One host synchronisation per render (the number of rays that enter the cube bounds the launches) plus one for the
hit count.  Hits are polished with one Newton step along the ray using the decoder's input Jacobian (sdfr_mlp_jacobian), which also
gives the normals and d sdf / d latent.
Gradients: the hit depth is an implicit function of pose and latent, f(o(θ) + λ d(θ), z(θ)) = 0, so
    λ(θ) = λ* - [ ∇f · (o(θ) + λ* d(θ) - x*) + ∂f/∂z · (z(θ) - z*) ] / (∇f · d*)
is evaluated with torch ops on the (N_hit, 3) tensors and autograd differentiates it -- exact first-order derivatives of depth, hit point
(NOCS colour) and normals' rotation w.r.t. yaw, trans and latent at fixed hit set (silhouette changes carry no gradient, as in the splat path).
"""
import torch
import torch.nn.functional as F

from .. import _lib
from ..deepsdf.networks.deep_sdf_decoder_scale import SdfState, mlp_jacobian


def _rot_from_yaw(yaw):
    c, s = torch.cos(yaw), torch.sin(yaw)
    z, o = torch.zeros_like(c), torch.ones_like(c)
    R = torch.stack([torch.stack([c, z, s], -1), torch.stack([z, o, z], -1), torch.stack([-s, z, c], -1)], -2)     # utils/refinement.py:108-125
    return R * torch.tensor([1.0, -1.0, 1.0], device=yaw.device).view(1, 3, 1)                                  # row 1 negated (optimizer.py:88)


class _CropRows(torch.autograd.Function):
    """x[b_idx] for per-crop rows x [B, ...] and an ASCENDING crop index per hit.  The backward is a sum over each crop's contiguous run of
    hits (one reduction per crop, deterministic) -- autograd's own backward of advanced indexing sorts and serialises the 18 k duplicates of
    a crop's index (5 ms per gathered tensor at one 256x256 crop: two thirds of the whole render)."""

    @staticmethod
    def forward(ctx, x, b_idx, bounds):
        ctx.bounds, ctx.shape = bounds, x.shape
        return x.index_select(0, b_idx)

    @staticmethod
    def backward(ctx, g):
        out = g.new_zeros(ctx.shape)
        for b, (lo, hi) in enumerate(ctx.bounds):
            if hi > lo:
                out[b] = g[lo:hi].sum(0)
        return out, None, None


class SphereTracer:
    def __init__(self, decoder, K, resolution_px, batch=1, steps=64, eps=2e-3, bound=1.0, relax=1.0, near=1e-3, device="cuda"):
        dev = torch.device(device)
        if dev.type != "cuda":
            raise _lib.SdfrError("SphereTracer runs on the GPU only")
        self.dev, self.B = dev, int(batch)
        self.W, self.H = int(resolution_px[0]), int(resolution_px[1])
        self.steps, self.eps, self.bound, self.relax, self.near = int(steps), float(eps), float(bound), float(relax), float(near)
        self.decoder = decoder
        self.handle = decoder.handle(dev)
        self.half = 1 if getattr(decoder, "mlp_precision", torch.float32) == torch.float16 else 0
        self.L = decoder.latent_size
        self.NI = self.L + 3
        K = torch.as_tensor(K, dtype=torch.float32)
        if K.dim() == 2:
            K = K.unsqueeze(0).expand(self.B, 3, 3)
        self.K = K.contiguous().to(dev)
        self.Kinv = torch.linalg.inv(K.cpu().float()).contiguous().to(dev)
        B, P = self.B, self.W * self.H
        f = lambda *s: torch.zeros(s, dtype=torch.float32, device=dev)
        i = lambda *s: torch.zeros(s, dtype=torch.int32, device=dev)
        self.counters = i(3)
        self.pix, self.lam = [i(B * P), i(B * P)], [f(B * P), f(B * P)]
        self.far, self.inputs, self.sdf = f(B * P), f(B * P, self.NI), f(B * P)
        self.hit_lam, self.hit_sdf = f(B * P), f(B * P)
        yy, xx = torch.meshgrid(torch.arange(self.H, device=dev), torch.arange(self.W, device=dev), indexing="ij")
        self.pixel_h = torch.stack([xx.reshape(-1), yy.reshape(-1), torch.ones(P, device=dev, dtype=torch.long)], -1).float()    # (P,3)
        self.check_every = 8            # host looks at the active count every 8 steps and stops the march when it is empty (0: never)
        self.stop_fraction = 5e-4       # ... or holds fewer than this fraction of the rays that entered the cube (they count as unresolved)
        self.steps_run = 0

    # ------------------------------------------------------------------------------------------------------------------
    def march(self, pose, latn):
        """the march itself (no autograd): fills hit_lam / hit_sdf [B*P]; returns the number of rays that entered the cube"""
        L = _lib.lib()
        P, ck = _lib.ptr, _lib.check
        B, W, H = self.B, self.W, self.H
        with _lib.guard(self.dev):
            st = _lib.stream_ptr()
            self.hit_lam.zero_(); self.hit_sdf.zero_()
            pose_c, latn_c = pose.contiguous(), latn.contiguous()
            ck(L.sdfr_trace_setup(P(pose_c), P(self.Kinv), P(latn_c), self.L, B, W, H, self.bound, self.near, P(self.counters), P(self.pix[0]),
                                  P(self.lam[0]), P(self.far), P(self.inputs), st), "sdfr_trace_setup")
            n0 = int(self.counters[0])                              # one synchronisation: bounds every launch of the march
            cptr = self.counters.data_ptr()
            import ctypes
            for s in range(self.steps):
                a, b = s & 1, (s + 1) & 1
                ck(L.sdfr_mlp_forward_counted(self.handle.h, P(self.inputs), n0, ctypes.c_void_p(cptr + 4 * (s % 3)), P(self.sdf), self.half, st),
                   "sdfr_mlp_forward_counted")
                ck(L.sdfr_trace_step(P(pose_c), P(self.Kinv), P(latn_c), self.L, W, H, self.eps, self.relax, P(self.sdf), P(self.counters), s, n0,
                                     P(self.pix[a]), P(self.lam[a]), P(self.pix[b]), P(self.lam[b]), P(self.far), P(self.inputs), P(self.hit_lam),
                                     P(self.hit_sdf), st), "sdfr_trace_step")
                done = s + 1
                # every few steps look at the active count: a step costs one decoder pass of latency even for a single ray (0.5 ms with the
                # f32 decoder), and after ~30 steps only a few rays creeping along the surface are left
                if self.check_every and done % self.check_every == 0 and done < self.steps and \
                        int(self.counters[done % 3]) <= self.stop_fraction * n0:
                    break
            self.steps_run = done
            self.n_entered = n0
            self.n_unresolved = self.counters[done % 3]             # device scalar: rays still active after the last step (treated as misses)
        return n0

    def forward(self, yaw, trans, latent, newton=True):
        B, W, H, P_ = self.B, self.W, self.H, self.W * self.H
        dev = self.dev
        yaw, trans, latent = yaw.reshape(B), trans.reshape(B, 3), latent.reshape(B, self.L)
        R = _rot_from_yaw(yaw)                                                       # (B,3,3), differentiable
        latn = F.normalize(latent, p=2, dim=1)                                       # optimizer.py:96
        with torch.no_grad():
            pose = torch.zeros(B, 4, 4, device=dev)
            pose[:, :3, :3] = R
            pose[:, :3, 3] = trans
            pose[:, 3, 3] = 1.0
            self.march(pose.view(B, 16), latn.detach())
            gp = torch.nonzero(self.hit_lam > 0).view(-1)                            # hit pixels (second synchronisation)
            nh = int(gp.numel())
        out = {"color": torch.zeros(B, 3, H, W, device=dev), "mask": torch.zeros(B, 1, H, W, device=dev),
               "depth": torch.zeros(B, 1, H, W, device=dev), "normals": torch.zeros(B, 3, H, W, device=dev)}
        self.n_hit = nh
        if nh == 0:
            return out
        b_idx, p_idx = gp // P_, gp % P_
        r_cam = (self.Kinv[b_idx] @ self.pixel_h[p_idx].unsqueeze(-1)).squeeze(-1)   # (nh,3) constants
        # hits are in ascending (crop, pixel) order: crop b owns the contiguous run bounds[b] (one small host read)
        edges = torch.searchsorted(b_idx, torch.arange(B + 1, device=dev)).tolist()
        bounds = list(zip(edges[:-1], edges[1:]))
        Rh, th, zh = _CropRows.apply(R, b_idx, bounds), _CropRows.apply(trans, b_idx, bounds), _CropRows.apply(latn, b_idx, bounds)
        d = torch.einsum("nij,ni->nj", Rh, r_cam)                                    # R^T r
        o = -torch.einsum("nij,ni->nj", Rh, th)                                      # -R^T t
        with torch.no_grad():
            lam0 = self.hit_lam[gp]
            x0 = o + lam0.unsqueeze(-1) * d
            rows = torch.cat([zh.detach(), x0], 1).contiguous()
            state = SdfState(self.handle, rows)
            state.f16 = False
            idx = torch.arange(nh, dtype=torch.int32, device=dev)
            J, f0 = mlp_jacobian(state, idx, nh, use_masks=False)                    # exact-f32 decoder value and input Jacobian at the hits
            gz, gx = J[:, :self.L], J[:, self.L:]
            gd = (gx * d).sum(-1)
            # polish only rays that meet the surface at more than ~6 degrees: along a grazing ray the first-order step is long and leaves the
            # linear region of the decoder (those hits keep the marched point, |sdf| < eps)
            ok = gd.abs() > 0.1 * gx.norm(dim=1) * d.norm(dim=1)
            lam_s = torch.where(ok, lam0 - f0 / torch.where(ok, gd, torch.ones_like(gd)), lam0) if newton else lam0
            x_s = o + lam_s.unsqueeze(-1) * d
            self.hit_residual = f0                                                   # decoder value at the marched points (diagnostic)
            n_hat = F.normalize(gx, dim=1)
            gd_s = torch.where(ok, gd, torch.ones_like(gd))
            z_s = zh.detach()
        # implicit-function reparametrisation: zero in value, exact first-order dependence on pose and latent
        f_lin = (gx * (o + lam_s.unsqueeze(-1) * d - x_s)).sum(-1) + (gz * (zh - z_s)).sum(-1)
        lam = lam_s - f_lin / gd_s
        x = o + lam.unsqueeze(-1) * d
        depth = lam * r_cam[:, 2]
        nocs = (x * torch.tensor([-1.0, 1.0, 1.0], device=dev) + 1) / 2              # projection.py:53-55, rasterer.py:113-114
        n_cam = torch.einsum("nij,nj->ni", Rh, n_hat)                                # R n (normals constant w.r.t. the latent, as grid.py:57-58)
        flat = b_idx * P_ + p_idx
        out["color"] = torch.zeros(B * P_, 3, device=dev).index_put((flat,), nocs).view(B, H, W, 3).permute(0, 3, 1, 2)
        out["normals"] = torch.zeros(B * P_, 3, device=dev).index_put((flat,), (n_cam + 1) / 2).view(B, H, W, 3).permute(0, 3, 1, 2)
        out["depth"] = torch.zeros(B * P_, device=dev).index_put((flat,), depth).view(B, 1, H, W)
        out["mask"] = torch.zeros(B * P_, device=dev).index_put((flat,), torch.ones(nh, device=dev)).view(B, 1, H, W)
        return out

    __call__ = forward
