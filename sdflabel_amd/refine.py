"""BatchRefiner -- the reference's whole refinement loop (pipelines/optimizer.py:43-237) for B crops, device resident.

The reference's `Optimizer.optimize` runs one crop at a time and pays, per iteration, an H2D upload of the lidar cloud (:84), a D2H
round trip plus a sklearn KDTree build for the 3-D loss (:180-181), Q x H x W dense tensors for the 2-D loss (:213-234) and several
`.item()` syncs (:149,154,188).  Here one iteration is ~25 kernel launches on one stream with no host synchronisation at all:

    BatchRenderer.forward  ->  sdfr_loss_2d (NOCS window loss)  +  sdfr_loss_3d (exact nearest neighbour loss)
                           ->  BatchRenderer.backward           ->  sdfr_solver_step (Adam on yaw/trans, SGD on scale/latent, skip rules)

with the same arithmetic per crop as the reference (trajectory-tested against its own Optimizer, golden G8).  Parameters live in ONE flat
structure-of-arrays buffer [ yaw(B) | trans(B,3) | scale(B) | latent(B,L) ] whose sections are the dense arrays the kernels read.
"""
import torch
import torch.nn.functional as F

from . import _lib
from .batch import BatchRenderer


class BatchRefiner:
    def __init__(self, decoder, density, K, crop_size, batch, lidar_cap, weights=None, cap=None, device="cuda", optimize_latent=True,
                 render="splat", trace_grad="surfel", tracer_kwargs=None, max_pixels=None, max_side=None, candidate_reuse=None):
        """crop_size = (H, W) as the reference passes it (optimizer.py:56,72 builds the Rasterer with crop_size[::-1]).
        optimize_latent=False: pose-only refinement (yaw, trans, scale; the latent parameter group of optimizer.py:38 gets no update), so
        the shape is evaluated once per set_crops() and every iteration only re-projects, splats and differentiates the pose.
        render="splat" (default): the reference's renderer (decoder on the grid -> surfels -> splat), reproduces its trajectories (G8*).
        render="trace": the sphere tracer (renderer/sphere_tracer.py; NOT the reference's algorithm, no parity claim) supplies what the loop
        reads from its renderer (optimizer.py:110-141): rendering['color'] = the traced NOCS image -> 2-D loss, points['xyzf'] = the
        camera-frame hit points in pixel order -> 3-D loss; its backward feeds the same solver step.  `density` is unused then.
        trace_grad="surfel" (default): hits differentiate as material points, the autograd semantics of the reference's surfels -- the mode
        the loop converges with; "image": image-space implicit-function gradients at the fixed pixels (DESIGN.md 3.6 has the comparison).
        tracer_kwargs: SphereTracer options (steps, cone_block, polish, ...).
        max_pixels / max_side (both renderers; r04): ragged extents -- every crop of a batch its own image size (H_b, W_b) and intrinsics K_b,
        given to set_crops(); buffers are sized for max_pixels pixels per crop and the captured graph serves every crop set within the caps
        (the reference pipeline's crops all differ: utils/refinement.py:586-609, pipelines/refine_css.py:117-129).  crop_size is then the
        default extent.  A crop refines bit-identically to the same crop alone in a fixed-size refiner.
        candidate_reuse (float16 decoders; None: decoder.candidate_reuse): evaluate the half decoder on the band candidates alone while a
        proven bound keeps them valid -- bit-identical results, 4x the crops/s (BatchRenderer)."""
        self.H, self.W = int(crop_size[0]), int(crop_size[1])
        self.B = int(batch)
        self._K0 = torch.as_tensor(K, dtype=torch.float32).detach().cpu().clone()
        self.w2 = float((weights or {}).get('2d', 0.3))          # configs/config_refine.ini:26-27
        self.w3 = float((weights or {}).get('3d', 0.5))
        self.optimize_latent = bool(optimize_latent)
        self.unresolved_last = 0            # render='trace': rays unresolved at the step budget in the last checked iteration (check_overflow)
        if render not in ("splat", "trace"):
            raise ValueError("render must be 'splat' or 'trace'")
        if trace_grad not in ("surfel", "image"):
            raise ValueError("trace_grad must be 'surfel' or 'image'")
        self.render, self.surfel = render, trace_grad == "surfel"
        B = self.B
        if render == "trace":
            from .renderer.sphere_tracer import SphereTracer
            self.br = None
            tracer_kwargs = dict(tracer_kwargs or {})
            # rays still marching at the step budget (1-2 grazing rays out of 65 k are normal: tests/test_gpu_sphere_tracer.py) count as misses;
            # check_overflow() raises only above max(2, unresolved_tolerance x pixels) of the last iteration and reports the count otherwise
            self.unresolved_tolerance = float(tracer_kwargs.pop("unresolved_tolerance", 1e-3))
            self.tr = SphereTracer(decoder, K, (self.W, self.H), batch, device=device, points=True, max_pixels=max_pixels, max_side=max_side,
                                   **tracer_kwargs)
            dev, self.L, est_cap = self.tr.dev, self.tr.L, self.tr.ecap
        else:
            self.tr = None
            self.br = BatchRenderer(decoder, density, K, (self.W, self.H), batch, cap=cap, device=device, max_pixels=max_pixels, max_side=max_side,
                                    candidate_reuse=candidate_reuse)
            self.br.freeze_shape = not self.optimize_latent
            dev, self.L, est_cap = self.br.dev, self.br.L, self.br.cap
        br = self.br
        self.dev = dev
        n = (5 + self.L) * B
        self.params = torch.zeros(n, dtype=torch.float32, device=dev)
        self.grads = torch.zeros(n, dtype=torch.float32, device=dev)

        def sections(buf):
            return buf[0:B], buf[B:4 * B].view(B, 3), buf[4 * B:5 * B], buf[5 * B:].view(B, self.L)

        self.yaw, self.trans, self.scale, self.latent = sections(self.params)
        self.g_yaw, self.g_trans, self.g_scale, self.g_latent = sections(self.grads)
        # the renderer reads / writes the sections of the flat buffers directly
        rd = br if br is not None else self.tr
        rd.yaw, rd.trans, rd.latent = self.yaw, self.trans, self.latent
        rd.g_yaw, rd.g_trans, rd.g_latent = self.g_yaw, self.g_trans, self.g_latent
        self.lidar_cap = int(lidar_cap)
        self.lidar = torch.zeros((B, self.lidar_cap, 3), dtype=torch.float32, device=dev)
        self.lcnt = torch.zeros((B,), dtype=torch.int32, device=dev)
        self.rd = br if br is not None else self.tr             # the renderer: BatchRenderer or SphereTracer (same extents interface)
        self.ragged = bool(self.rd.ragged)
        img_shape = (B, 3, self.rd.PS) if self.ragged else (B, 3, self.H, self.W)
        self.target = torch.zeros(img_shape, dtype=torch.float32, device=dev)
        self.loss2d = torch.zeros((B,), dtype=torch.float32, device=dev)
        self.loss3d = torch.zeros((B,), dtype=torch.float32, device=dev)
        self.total = torch.zeros((B,), dtype=torch.float32, device=dev)
        self.nvalid = torch.zeros((B,), dtype=torch.int32, device=dev)
        self.npairs = torch.zeros((B,), dtype=torch.int32, device=dev)
        self.stepped = torch.zeros((B,), dtype=torch.int32, device=dev)
        self.g_color = torch.zeros(img_shape, dtype=torch.float32, device=dev)
        self.l2_scratch = torch.zeros((3 * B * (self.rd.tiles16_cap if self.ragged else ((self.W + 15) // 16) * ((self.H + 15) // 16)),), dtype=torch.float32, device=dev)
        self.l3_scratch = torch.zeros((3 * B * ((est_cap + 63) // 64),), dtype=torch.float32, device=dev)
        self.g_xyzf = torch.zeros((B, est_cap, 3), dtype=torch.float32, device=dev)
        self.adam_m = torch.zeros((B, 4), dtype=torch.float32, device=dev)
        self.adam_v = torch.zeros((B, 4), dtype=torch.float32, device=dev)
        self.adam_t = torch.zeros((B,), dtype=torch.int32, device=dev)
        # r06: both losses in one launch, the backward tail and the solver step in one launch (same bits; decoder.fused_launches = False keeps
        # the r05 sequence).  kscale[b] = (2-D, 3-D) normalisation factors the consumers multiply on load.
        self.fused = br is not None and br.fused and self.L <= 8
        self.kscale = torch.zeros((B, 2), dtype=torch.float32, device=dev)
        self._replay = None

    # ------------------------------------------------------------------------------------------------------------------
    def set_crops(self, params, nocs_pred, lidars, K=None, crop_sizes=None):
        """params: dict of (B, .) arrays 'yaw' (B,1|B), 'trans' (B,3), 'scale' (B,1|B), 'latent' (B,L)  (optimizer.py:26-40);
        nocs_pred: (B,3,h,w) CSS-net NOCS predictions -- or, with ragged extents, a list of B (3,h_b,w_b) predictions -- resized with
        nearest-neighbour interpolation to each crop's size (optimizer.py:135-137);
        lidars: list of B (M_b,3) arrays (camera-frame lidar points of each crop's frustum);
        K (B,3,3) | (3,3) and crop_sizes [(H_b, W_b)] * B (ragged extents only): the crops' own intrinsics and image sizes as the reference
        passes them to Optimizer.optimize (refine_css.py:203-223); default: those given at construction."""
        B, dev = self.B, self.dev
        t = lambda a: torch.as_tensor(a, dtype=torch.float32, device=dev)
        if (K is not None or crop_sizes is not None) and not self.ragged:
            raise _lib.SdfrError("per-crop K / crop_sizes need a BatchRefiner built with max_pixels (ragged extents)")
        self.yaw.copy_(t(params['yaw']).reshape(B))
        self.trans.copy_(t(params['trans']).reshape(B, 3))
        self.scale.copy_(t(params['scale']).reshape(B))
        self.latent.copy_(t(params['latent']).reshape(B, self.L))
        if self.ragged:
            sizes = [(int(h), int(w)) for h, w in crop_sizes] if crop_sizes is not None else [(self.H, self.W)] * B
            # K=None means the constructor's intrinsics, as the docstring says -- not whatever the previous crop set left in place
            self.rd.set_extents([(w, h) for h, w in sizes], K if K is not None else self._K0)
            self.target.zero_()
            for b, (h, w) in enumerate(sizes):
                pred = t(nocs_pred[b])
                self.target[b, :, :h * w] = F.interpolate(pred[None], size=(h, w), mode='nearest')[0].reshape(3, h * w)
        else:
            self.target.copy_(F.interpolate(t(nocs_pred), size=(self.H, self.W), mode='nearest'))
        self.lidar.zero_()
        for b, l in enumerate(lidars):
            l = t(l).reshape(-1, 3)
            if l.shape[0] > self.lidar_cap:
                raise _lib.SdfrError("crop %d has %d lidar points > lidar_cap %d" % (b, l.shape[0], self.lidar_cap))
            self.lidar[b, :l.shape[0]] = l
            self.lcnt[b] = l.shape[0]
        self.adam_m.zero_(); self.adam_v.zero_(); self.adam_t.zero_()
        if self.br is not None:
            self.br.invalidate_shape()
            self.br.clear_overflow()            # sticky truncation flags belong to the crops refined before
            self.br.reset_guard()               # per-crop device state of the two-stage mode (violation counters, margins) starts clean
        # a captured graph stays valid: every buffer it reads or writes is static and was updated in place above

    def iteration(self):
        """One refinement iteration of every crop (optimizer.py:79-157).  No host synchronisation."""
        with _lib.guard(self.dev):
            self._iteration()

    def _iteration(self):
        L = _lib.lib()
        P, st, ck = _lib.ptr, _lib.stream_ptr(), _lib.check
        br, B = self.br, self.B
        if br is None:
            self._iteration_traced(L, P, st, ck)
            return
        out = br.forward()
        if self.fused:
            ck(L.sdfr_losses_fused(P(out["color"]), P(self.target), B, self.H, self.W, P(br.wh) if self.ragged else None, br.PS,
                                   br.tiles16_cap if self.ragged else 0, 5.0, 1.0, self.w2, P(self.loss2d), P(self.g_color), P(self.nvalid),
                                   P(self.l2_scratch), P(out["xyzf"]), P(br.fcnt), br.cap, P(self.lidar), P(self.lcnt), self.lidar_cap, P(self.scale),
                                   0.2, self.w3, P(self.loss3d), P(self.g_xyzf), P(self.g_scale), P(self.npairs), P(self.l3_scratch), P(self.kscale),
                                   st), "sdfr_losses_fused")
            br.backward_solve(self.g_color, self.g_xyzf, self.kscale,
                              {"params": self.params, "grads": self.grads, "loss2d": self.loss2d, "loss3d": self.loss3d, "npairs": self.npairs,
                               "w2": self.w2, "w3": self.w3, "adam_m": self.adam_m, "adam_v": self.adam_v, "adam_t": self.adam_t,
                               "lr_latent": 0.00003 if self.optimize_latent else 0.0, "total": self.total, "stepped": self.stepped})
            return
        if self.ragged:
            ck(L.sdfr_loss_2d_r(P(out["color"]), P(self.target), B, P(br.wh), br.PS, br.tiles16_cap, 5.0, 1.0, self.w2, P(self.loss2d),
                                P(self.g_color), P(self.nvalid), P(self.l2_scratch), st), "sdfr_loss_2d_r")
        else:
            ck(L.sdfr_loss_2d(P(out["color"]), P(self.target), B, self.H, self.W, 5.0, 1.0, self.w2, P(self.loss2d), P(self.g_color),
                              P(self.nvalid), P(self.l2_scratch), st), "sdfr_loss_2d")
        ck(L.sdfr_loss_3d(P(out["xyzf"]), P(br.fcnt), br.cap, P(self.lidar), P(self.lcnt), self.lidar_cap, P(self.scale), 0.2, self.w3, B,
                          P(self.loss3d), P(self.g_xyzf), P(self.g_scale), P(self.npairs), P(self.l3_scratch), st), "sdfr_loss_3d")
        br.backward(g_color=self.g_color, g_xyzf=self.g_xyzf)
        if not self.optimize_latent:
            self.g_latent.zero_()                       # no latent parameter group: exactly no update (lr * 0)
        ck(L.sdfr_solver_step(P(self.params), P(self.grads), self.L, P(self.loss2d), P(self.loss3d), P(self.npairs), self.w2, self.w3,
                              P(self.adam_m), P(self.adam_v), P(self.adam_t), 0.01, 0.01, 0.00003 if self.optimize_latent else 0.0, B,
                              P(self.total), P(self.stepped), st), "sdfr_solver_step")

    def _iteration_traced(self, L, P, st, ck):
        """the same iteration with the sphere tracer as the loop's renderer (optimizer.py:110-123 -> rendering['color'], points['xyzf'])"""
        tr, B = self.tr, self.B
        out = tr.render()
        if self.ragged:
            ck(L.sdfr_loss_2d_r(P(out["color"]), P(self.target), B, P(tr.wh), tr.PS, tr.tiles16_cap, 5.0, 1.0, self.w2, P(self.loss2d), P(self.g_color),
                                P(self.nvalid), P(self.l2_scratch), st), "sdfr_loss_2d_r")
        else:
            ck(L.sdfr_loss_2d(P(out["color"]), P(self.target), B, self.H, self.W, 5.0, 1.0, self.w2, P(self.loss2d), P(self.g_color),
                              P(self.nvalid), P(self.l2_scratch), st), "sdfr_loss_2d")
        ck(L.sdfr_loss_3d(P(out["xyzf"]), P(tr.ecnt), tr.ecap, P(self.lidar), P(self.lcnt), self.lidar_cap, P(self.scale), 0.2, self.w3, B,
                          P(self.loss3d), P(self.g_xyzf), P(self.g_scale), P(self.npairs), P(self.l3_scratch), st), "sdfr_loss_3d")
        tr.backward(g_color=self.g_color, g_xyzf=self.g_xyzf, surfel=self.surfel)
        if not self.optimize_latent:
            self.g_latent.zero_()
        ck(L.sdfr_solver_step(P(self.params), P(self.grads), self.L, P(self.loss2d), P(self.loss3d), P(self.npairs), self.w2, self.w3,
                              P(self.adam_m), P(self.adam_v), P(self.adam_t), 0.01, 0.01, 0.00003 if self.optimize_latent else 0.0, B,
                              P(self.total), P(self.stepped), st), "sdfr_solver_step")

    def capture(self):
        """Capture one iteration in a HIP graph; optimize() then replays it."""
        s = torch.cuda.Stream(device=self.dev)
        s.wait_stream(torch.cuda.current_stream(self.dev))
        snap = (self.params.clone(), self.adam_m.clone(), self.adam_v.clone(), self.adam_t.clone())
        br = self.br
        guard = (br.violations.clone(), br.margin_dev.clone(), br.max_dev.clone()) if (br is not None and br.guarded) else None
        with torch.cuda.stream(s):
            self.iteration()
        torch.cuda.current_stream(self.dev).wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self.iteration()
        # warm-up and capture must not advance the optimisation
        self.params.copy_(snap[0]); self.adam_m.copy_(snap[1]); self.adam_v.copy_(snap[2]); self.adam_t.copy_(snap[3])
        if guard is not None:                   # ... nor feed the two-stage mode's guard counters
            br.violations.copy_(guard[0]); br.margin_dev.copy_(guard[1]); br.max_dev.copy_(guard[2]); br.age.zero_()
        self._replay = g.replay
        self.captures = getattr(self, "captures", 0) + 1          # (bench / tests: how often this refiner had to capture)
        return g.replay

    def optimize(self, iters_optim):
        for _ in range(iters_optim):
            if self._replay is not None and self.br is not None and self.br.freeze_shape and not self.br._shape_valid:
                self.iteration()                        # pose-only: the captured graph holds the steady state; a new latent needs one full pass
                continue
            if self._replay is not None:
                self._replay()
            else:
                self.iteration()

    def results(self):
        """(B, 5+L) rows [yaw, trans(3), scale, latent(L)] and the last (B,) weighted losses (2d, 3d).  Synchronises; raises if a crop's
        band overflowed the surfel capacity in the last iteration (its shape would have been truncated)."""
        self.check_overflow()
        rows = torch.cat([self.yaw.view(-1, 1), self.trans, self.scale.view(-1, 1), self.latent], dim=1)
        return rows.clone(), (self.w2 * self.loss2d).clone(), (self.w3 * self.loss3d).clone()

    def check_overflow(self):
        """splat: raise if a crop's band exceeded the surfel capacity (BatchRenderer.check_overflow).  trace: the point list holds every pixel,
        so nothing overflows, but rays that exhausted the march's step budget in the last iteration raise.  One synchronisation."""
        if self.br is not None:
            self.br.check_overflow()
        else:
            # rays still marching when the step budget ran out count as misses: above the tolerance the last iteration refined against an
            # incomplete hit set (too few steps for this view) -- the traced counterpart of a truncated band.  A couple of grazing
            # rays creeping along the surface are normal and only reported: self.unresolved_last, results()' callers may read it.
            n = self.unresolved_last = int(self.tr.n_unresolved)
            pixels = sum(w * h for w, h in self.tr.sizes) if self.ragged else self.B * self.H * self.W
            if n > max(2, self.unresolved_tolerance * pixels):
                raise _lib.SdfrError("sphere tracer: %d ray(s) of %d were unresolved after %d steps in the last iteration (treated as misses; "
                                     "tolerance max(2, %g x pixels)): raise tracer_kwargs['steps'] or tracer_kwargs['unresolved_tolerance']"
                                     % (n, pixels, self.tr.steps, self.unresolved_tolerance))
