"""BatchRenderer -- the whole renderer hot path for B crops per call, device-resident and free of host synchronisation.

The reference refines one crop at a time (pipelines/refine_css.py:94) and pays, per iteration, dozens of tiny ATen launches and
three host syncs (pipelines/optimizer.py:79-164).  This class is the MI355X-native extension SURVEY.md §8(f3) asks for: it keeps
the reference's arithmetic per crop (same kernels as the drop-in modules, parity-tested against them) but takes the optimizer's
parameters directly,

    forward (yaw[B], trans[B,3], latent[B,L])  ->  color[B,3,H,W], mask[B,1,H,W], depth[B,1,H,W], normals[B,3,H,W],
                                                    xyzf[B,cap,3] (front-facing camera-frame points, zero padded), nf[B], n[B]
    backward(g_color, g_mask, g_depth, g_normals, g_xyzf)  ->  g_yaw[B], g_trans[B,3], g_latent[B,L]

with every buffer pre-allocated (ragged per-crop data lives in [B][cap] arrays with device-side counts), so a step is ~16 kernel
launches on the current stream and can be captured in a HIP graph (`capture()`).  `overflow()` (one sync, call it when convenient)
reports whether any crop's band exceeded `cap`; `check_overflow()` raises (the refinement loop calls it once after the iterations).

Modes (all leave the arithmetic of what is consumed downstream unchanged):
  binned          from 4 crops per launch the splat uses per-tile surfel lists built inside sdfr_surfels_forward (same bits as the scan)
  freeze_shape    pose-only refinement: decoder, band and Jacobian evaluated once per latent, later forwards only re-project and splat
  decoder.mlp_precision  float32 (exact-f32 MFMA, the parity path) | float16 | "float32_split" | "float32_prefilter" (two-stage evaluation
                  with a device-side run-time guard; decoder.prefilter_reuse additionally skips the half pass while the candidate set is
                  provably still valid) -- DESIGN.md 3.2
  decoder.candidate_reuse / BatchRenderer(candidate_reuse=True)  (r05; float32 and float16) the mode's own kernel on the band candidates alone
                  while a proven Lipschitz bound keeps the candidate set valid: bit-identical to evaluating the whole grid, 4-8x the crops/s
"""
import torch

from . import _lib
from .grid import Grid3D

_DIAM_DISC = 0.04
_DEPTH_CONSTANT = 150.0


class BatchRenderer:
    def __init__(self, decoder, density, K, resolution_px, batch, cap=None, device="cuda", threshold=0.03, output_nocs=True,
                 max_pixels=None, max_side=None, candidate_reuse=None):
        """max_pixels (r04, ragged extents): every crop of the batch may have its OWN image size (W_b, H_b) with W_b H_b <= max_pixels and
        W_b, H_b <= max_side (default 4 sqrt(max_pixels)), and its own intrinsics -- what the reference pipeline's crops look like
        (utils/refinement.py:586-609).  Images then live in slots of max_pixels pixels per channel ([B, C, max_pixels]; image(b, name) gives
        the (C, H_b, W_b) view), the extents sit on the device (set_extents) and the kernels read them there: one set of buffers and ONE
        captured HIP graph serve any crop sizes within the caps.  A crop's results are bit-identical to rendering it alone at its size.
        resolution_px is then the initial extent of every crop."""
        dev = torch.device(device)
        if dev.type != "cuda":
            raise _lib.SdfrError("BatchRenderer runs on the GPU only")
        self.dev = dev
        self.B = int(batch)
        self.W, self.H = int(resolution_px[0]), int(resolution_px[1])
        self.thr = float(threshold)
        self.nocs_mode = 1 if output_nocs else 0
        if not output_nocs:
            raise NotImplementedError("BatchRenderer composites NOCS colours (the optimizer's configuration)")
        self.decoder = decoder
        prec = getattr(decoder, "mlp_precision", torch.float32)
        self.f16 = prec == torch.float16
        self.split = prec == "float32_split"
        # two-stage evaluation: half-operand pass over the grid -> candidates |sdf| < threshold + margin -> exact float32 pass (sdf + Jacobian)
        # on the candidates only -> exact band.  The margin must exceed the half pass's error (4e-4 on the shipped decoder).
        self.prefilter = prec == "float32_prefilter"
        self.margin = float(getattr(decoder, "prefilter_margin", 0.005))
        self.handle = decoder.handle(dev)
        self.fused = bool(getattr(decoder, "fused_launches", True))      # r06 (below): fused launches, same bits; False keeps the r05 sequence
        self.L = decoder.latent_size
        self.NI = self.L + 3
        self.grid = Grid3D(density, dev).points.detach().contiguous()
        self.G = self.grid.shape[0]
        self.cap = int(cap) if cap is not None else max(256, self.G // 8)
        B, G, cap, H, W, NI = self.B, self.G, self.cap, self.H, self.W, self.NI
        self.ragged = max_pixels is not None
        if self.ragged:
            import math
            self.PS = int(max_pixels)
            self.max_side = int(max_side) if max_side is not None else min(self.PS, 4 * int(math.ceil(math.sqrt(self.PS))))
            if W * H > self.PS or max(W, H) > self.max_side:
                raise _lib.SdfrError("resolution_px %dx%d exceeds max_pixels %d / max_side %d" % (W, H, self.PS, self.max_side))
            # tile counts any admissible shape can reach: ceil(W/t) ceil(H/t) <= W H / t^2 + (W + H) / t + 1
            self.tiles_cap = (self.PS + 63) // 64 + (2 * self.max_side + 7) // 8 + 2
            self.tiles16_cap = (self.PS + 255) // 256 + (2 * self.max_side + 15) // 16 + 2
            self.wh = torch.tensor([[W, H]] * B, dtype=torch.int32, device=dev)
            self.sizes = [(W, H)] * B
        else:
            self.PS = W * H
        K = torch.as_tensor(K, dtype=torch.float32)
        if K.dim() == 2:
            K = K.unsqueeze(0).expand(B, 3, 3)
        self.K = K.contiguous().to(dev)
        self.Kinv = torch.linalg.inv(K.cpu().float()).contiguous().to(dev)      # primitives.py:204, once, on the host

        def f(*shape):
            return torch.zeros(shape, dtype=torch.float32, device=dev)

        def i(*shape):
            return torch.zeros(shape, dtype=torch.int32, device=dev)

        # parameters (static addresses so that a captured graph can be replayed after in-place updates)
        self.yaw, self.trans, self.latent = f(B), f(B, 3), f(B, self.L)
        self.inputs, self.pose, self.latnorm = f(B * G, NI), f(B, 16), f(B)
        self.sdf = f(B * G)
        self.mask_ws = i(int(_lib.lib().sdfr_decoder_mask_words(self.handle.h, B * G)))
        self.idx, self.cnt, self.scratch = i(B, cap), i(B), i(B * ((G + 255) // 256) + 1)
        self.J, self.sdf_band = f(B, cap, NI), f(B, cap)
        if self.prefilter:
            self.cidx, self.ccnt, self.cslot = i(B, cap), i(B), i(B * G)
            self.Jc = f(B, cap, NI)
            # calibrate the margin on this decoder: the half pass against the exact pass on the grid for a few unit latents (the optimizer
            # normalises the latent, optimizer.py:96); the margin is at least 4x the largest deviation seen
            Lh = _lib.lib()
            gen = torch.Generator().manual_seed(0)
            s32, s16, s32b = f(G), f(G), f(G)
            worst, lip = 0.0, 0.0
            for _ in range(4):
                lat = torch.nn.functional.normalize(torch.randn(self.L, generator=gen), dim=0).to(dev)
                inp = torch.cat([lat.expand(G, -1), self.grid], 1).contiguous()
                _lib.check(Lh.sdfr_mlp_forward(self.handle.h, _lib.ptr(inp), G, _lib.ptr(s32), None, _lib.stream_ptr()), "sdfr_mlp_forward")
                _lib.check(Lh.sdfr_mlp_forward_f16(self.handle.h, _lib.ptr(inp), G, _lib.ptr(s16), None, _lib.stream_ptr()), "sdfr_mlp_forward_f16")
                worst = max(worst, float((s32 - s16).abs().max()))
                # Lipschitz constant of the decoder output in the normalised latent (candidate-set reuse): largest change of sdf on the grid
                # per unit of latent movement, for a small move along the unit sphere
                lat2 = torch.nn.functional.normalize(lat + 0.02 * torch.randn(self.L, generator=gen).to(dev), dim=0)
                inp2 = torch.cat([lat2.expand(G, -1), self.grid], 1).contiguous()
                _lib.check(Lh.sdfr_mlp_forward(self.handle.h, _lib.ptr(inp2), G, _lib.ptr(s32b), None, _lib.stream_ptr()), "sdfr_mlp_forward")
                lip = max(lip, float((s32b - s32).abs().max()) / max(float((lat2 - lat).norm()), 1e-12))
            self.f16_error = worst
            self.margin = max(self.margin, 4.0 * worst)
            # run-time guard (sdfr_prefilter_guard): the margin lives on the device, per crop; every step measures the half pass's deviation
            # at the candidates, grows a crop's margin to 4x the deviation when it exceeds half of it, and counts such steps
            self.margin_dev = torch.full((B,), self.margin, dtype=torch.float32, device=dev)
            self.max_dev = f(B)
            self.violations = i(B, 2)           # per crop, SINCE THE LAST reset_guard() (set_params / set_crops): [soft, hard]
            # candidate-set reuse (opt-in: decoder.prefilter_reuse = True): while the normalised latent has moved less than margin / (4 lip)
            # since the last half pass, that pass and the candidate selection are skipped (sdfr_prefilter_plan decides per crop on the device)
            # NOTE: `lip` is the largest finite difference of the decoder output seen over four random latent moves, times a
            # safety factor 4 -- a calibrated ESTIMATE, not a proven Lipschitz bound.  On reused steps the half pass does not run, so the guard
            # has nothing to compare and an underestimated constant could let a band row slip out of the candidate set unnoticed: reuse is an
            # approximation by design (opt-in; bit-identical to the plain two-stage mode in every test and in the 1024-crop bench run).
            # r05: the constant the plan kernel uses is the PROVEN bound (Decoder.latent_lipschitz_bound: spectral norms of the effective
            # weights along the latent's paths; r02-r04 used 4x the sampled finite difference, kept below as a diagnostic only -- the bound is
            # ~100x it on the shipped decoder, and the latent moves ~1e-6 per iteration, so reuse still covers most steps)
            self.lipschitz_sampled = lip
            self.lipschitz = float(decoder.latent_lipschitz_bound())
            self.reuse = bool(getattr(decoder, "prefilter_reuse", False))
            self.max_reuse = int(getattr(decoder, "prefilter_max_reuse", 16))
            self.lat_ref, self.age, self.reuse_flag = f(B, self.L), i(B), i(B)
            # audit (r04): the guard sees the half pass only at the candidates; every step a rotating 1 / audit_stride slice of the NON-candidate
            # rows is evaluated with the exact-f32 decoder as well, and a row that belongs to the band although it was never proposed counts a
            # hard violation like the guard's (check_overflow raises).  decoder.prefilter_audit = False turns it off (the r03 behaviour).
            self.audit = bool(getattr(decoder, "prefilter_audit", True))
            self.audit_stride = int(getattr(decoder, "prefilter_audit_stride", 16))
            # arithmetic of the audit's reference values: "split" (default) = float32-grade values from error-compensated f16 matrix products
            # (sdfr_mlp_forward_split: within 2.4e-7 of the exact-f32 kernel, 2.5x its speed -- the audit compares against a margin of ~1e-3);
            # "float32" = the exact-f32 kernel (r04's first version: 6.6 ms of a 21.6 ms step at 64 crops; split: see profiles/r04_notes.md section 9)
            self.audit_split = str(getattr(decoder, "prefilter_audit_arith", "split")) == "split" and self.handle.hp == 512 and not self.handle.has_ln
            if self.audit:
                self.audit_cap = B * ((G + self.audit_stride - 1) // self.audit_stride)
                self.audit_rows, self.audit_src, self.audit_sdf = f(self.audit_cap, NI), i(self.audit_cap), f(self.audit_cap)
                self.audit_n, self.audit_phase, self.audit_dev = i(1), i(1), f(B)
            self.fault = None           # tests: (flat grid rows int64 tensor, values) written over the half pass's output -- a planted half-pass error
        # float16 candidate reuse (r05, opt-in: decoder.candidate_reuse = True): the half decoder runs over the whole grid only when a crop's
        # candidate set (|sdf| < threshold + margin at that pass) may have gone stale; every other step evaluates the candidates alone, with the
        # same kernel, so band, values and Jacobian have the bits of the full-grid evaluation (csrc/surface.hip).  "May have
        # gone stale" is decided per crop on the device (sdfr_prefilter_plan) from a PROVEN bound: a row outside the candidates had
        # |h(z0)| >= thr + margin, and |h(z1) - h(z0)| <= lip |z1 - z0| + 2 e16, lip = Decoder.latent_lipschitz_bound() (product of the
        # spectral norms of the effective weights along the latent's paths: cannot be low), e16 = the half kernel's deviation from the exact
        # decoder (calibrated below; the margin is at least 4 e16).  Reuse while lip |z1 - z0| <= 0.45 margin.  On top, every step a rotating
        # 1 / audit_stride slice of the rows outside the candidates is evaluated too: one of them inside the band is a hard violation.
        want_reuse = bool(getattr(decoder, "candidate_reuse", False)) if candidate_reuse is None else bool(candidate_reuse)
        # ... for the float16 decoder AND for the exact-float32 one (prec == torch.float32: the parity path -- the same scheme with the f32 kernels)
        self.creuse = (self.f16 or prec == torch.float32) and want_reuse and self.handle.hp == 512 and not self.handle.has_ln
        self.reuse_off_reason = None
        if want_reuse and not self.creuse:
            self.reuse_off_reason = "candidate reuse needs a float16 / float32 decoder of padded width 512 without LayerNorm"
        if self.creuse:
            Lh = _lib.lib()
            # Kernel errors for the proof's budget (the guarantee is "PROVEN Lipschitz bound + CALIBRATED kernel errors with a safety
            # factor + run-time audit", not a proof end to end).
            #   E32: the exact-f32 kernel against the decoder in exact arithmetic, MEASURED per decoder (Decoder.kernel_error_f32: 1024 rows against
            #        a float64 evaluation on the host, cached per parameter set; 1.6e-7 on the shipped decoder), times 4, at least 1e-6.
            #   e16: the half kernel's deviation from the exact-f32 kernel on the grid for four unit latents (2.6e-4 on the shipped decoder), times
            #        decoder.candidate_error_safety (default 2: the sample maximum over 4 latents is not a bound), + E32.
            safety = float(getattr(decoder, "candidate_error_safety", 2.0))
            E32 = max(1e-6, 4.0 * float(decoder.kernel_error_f32(dev)))
            gen = torch.Generator().manual_seed(0)
            s32, s16 = f(G), f(G)
            dev16 = 0.0
            for _ in range(4):
                lat = torch.nn.functional.normalize(torch.randn(self.L, generator=gen), dim=0).to(dev)
                inp = torch.cat([lat.expand(G, -1), self.grid], 1).contiguous()
                _lib.check(Lh.sdfr_mlp_forward(self.handle.h, _lib.ptr(inp), G, _lib.ptr(s32), None, _lib.stream_ptr()), "sdfr_mlp_forward")
                _lib.check(Lh.sdfr_mlp_forward_f16(self.handle.h, _lib.ptr(inp), G, _lib.ptr(s16), None, _lib.stream_ptr()), "sdfr_mlp_forward_f16")
                dev16 = max(dev16, float((s32 - s16).abs().max()))
            self.calib_inputs = inp                # (tests: the last calibration rows, to check E32 against float64)
            self.f16_deviation_sampled, self.e32 = dev16, E32
            dev16 = safety * dev16
            # exact-f32 mode: the FULL-GRID pass only selects candidates -- every value consumed downstream comes from the exact kernel on the
            # candidates -- so it may run in half (decoder.candidate_select = "float16", the default; "float32": the exact kernel).  A row it
            # leaves out had |half value| >= thr + margin, i.e. |exact value| >= thr + margin - e_sel; margin >= 4 e_sel keeps the proof's budget
            # (0.45 margin for the latent, 0.25 for e_sel, 3 E32 ~ 0)
            self.select_half = (not self.f16) and str(getattr(decoder, "candidate_select", "float16")) == "float16"
            self.f16_error = (dev16 + E32) if self.f16 else E32          # the mode's own kernel against exact arithmetic
            self.select_error = (dev16 + E32) if self.select_half else self.f16_error
            need = 4.0 * max(self.f16_error, self.select_error)
            # a decoder whose kernel error needs a margin beyond decoder.candidate_max_margin (default: the band threshold itself -- the candidate
            # set would be more than twice the band) gets NO reuse: every step evaluates the whole grid, as the reference does
            max_margin = float(getattr(decoder, "candidate_max_margin", self.thr))
            lipschitz = float(decoder.latent_lipschitz_bound())
            if need > max_margin:
                self.creuse = False
                self.reuse_off_reason = ("kernel error %.3g (x%g safety) needs a candidate margin %.3g > candidate_max_margin %.3g"
                                         % (max(self.f16_error, self.select_error), safety, need, max_margin))
            elif not (lipschitz < float("inf")):
                self.creuse = False
                self.reuse_off_reason = "no finite latent Lipschitz bound for this decoder"
            del s32, s16
        if self.creuse:
            self.cstride = (cap + 127) // 128 * 128
            cs = self.cstride
            self.cidx, self.ccnt, self.cslot, self.cpos = i(B, cs), i(B), i(B * G), i(B, cap)
            self.crow, self.csdf = f(B * cs, NI), f(B * cs)
            self.cmask = i(int(Lh.sdfr_decoder_mask_words(self.handle.h, B * cs)))
            self.margin_grown = need > self.margin      # (prefilter_report: the calibrated error asked for more than decoder.prefilter_margin)
            self.margin = max(self.margin, need)
            self.margin_dev = torch.full((B,), self.margin, dtype=torch.float32, device=dev)
            self.max_dev = f(B)                 # (stays 0: this mode has no second arithmetic to deviate from; the plan kernel reads it)
            self.violations = i(B, 2)
            self.lipschitz = lipschitz
            # the plan kernel reuses while  lip_plan |z1 - z0| <= margin / 4.  A row outside the candidates had |h_sel(z0)| >= thr + margin at the
            # selecting pass; with F the decoder in exact arithmetic (|F(z1) - F(z0)| <= Lip |z1 - z0|) the value consumed at z1 is
            #   float16:        |h16(z1)| >= thr + margin - Lip |dz| - 2 e16
            #   exact float32:  |h32(z1)| >= thr + margin - Lip |dz| - e_sel - e32
            # so the latent may use  margin - (the kernel errors) ; 5 % of the margin stays unspent.  r05 gave the latent a flat 0.45 margin (the
            # worst case margin = 4 e16); with the calibrated errors of the shipped decoder the share is 0.74 (float16) / 0.85 (float32): candidate
            # sets stay valid 1.6-1.9x longer for the same proof.  The kernel is handed the bound scaled by 0.25 / share.
            kernel_errors = 2.0 * self.f16_error if self.f16 else self.select_error + self.f16_error
            self.latent_share = max(0.45, 0.95 - kernel_errors / self.margin)      # (>= 0.45 by calibration: margin >= 4 x the largest error)
            self.lipschitz_plan = self.lipschitz * (0.25 / self.latent_share)
            self.reuse = True
            # (the bound is proven, so no full pass is forced for safety's sake inside a 60-iteration refinement, configs/config_refine.ini:15;
            # measured at 64 crops per launch: max_reuse 16 -> 64 and audit stride 16 -> 32 take a refinement iteration from 4.5 to 3.5 ms)
            self.max_reuse = int(getattr(decoder, "candidate_max_reuse", 64))
            self.lat_ref, self.age, self.reuse_flag = f(B, self.L), i(B), i(B)
            self.audit = bool(getattr(decoder, "candidate_audit", True))
            self.audit_stride = int(getattr(decoder, "candidate_audit_stride", 32))
            self.audit_split = (not self.f16) and str(getattr(decoder, "candidate_audit_arith", "split")) == "split"
            self.audit_side = self.audit and B <= int(getattr(decoder, "candidate_audit_side_max_crops", 64)) and bool(getattr(decoder, "candidate_audit_side_stream", True))
            self._side = torch.cuda.Stream(device=dev) if self.audit_side else None
            self._side_pending = False
            self.half_tiles = self.f16 and B <= 2 and bool(getattr(decoder, "candidate_half_tiles", True))      # (a float16 option)
            # r06: 32-row tiles at ONE crop per launch (fused launches only: the pool kernel of sdfr_mlp_forward_candidates)
            self.quarter_tiles = self.half_tiles and B == 1 and self.fused and bool(getattr(decoder, "candidate_quarter_tiles", True))
            if self.audit:
                self.audit_cap = B * ((G + self.audit_stride - 1) // self.audit_stride)
                self.audit_rows, self.audit_src, self.audit_sdf = f(self.audit_cap, NI), i(self.audit_cap), f(self.audit_cap)
                self.audit_n, self.audit_phase, self.audit_dev = i(1), i(1), f(B)
            self.fault = None                   # tests: (flat grid rows, values) written over the full pass's output
        # r06: fused launches (same bits as the launch sequence they replace; False keeps the r05 sequence for A/B tests) and STICKY truncation
        # flags: over[b] bit 0 = the band exceeded cap in SOME forward since the last clear, bit 1 = the candidates exceeded their stride
        self.over = i(B)
        self.guarded = self.prefilter or self.creuse        # modes with device-side guard state (violations / margin / age)
        self.n_full = i(B) if self.guarded else None        # full-grid half passes per crop since reset_guard() (counted by the plan kernel)
        self.points, self.nocs, self.normals = f(B, cap, 3), f(B, cap, 3), f(B, cap, 3)
        self.p_cam, self.n_cam, self.attr = f(B, cap, 3), f(B, cap, 3), f(B, cap, 3)
        self.fidx, self.fcnt, self.fslot = i(B, cap), i(B), i(B, cap)
        if self.ragged:
            self.bbox = torch.empty((int(_lib.lib().sdfr_splat_ws_words_r(B, max(cap, 1), self.tiles_cap)),), dtype=torch.int32, device=dev)
        else:
            self.bbox = _lib.splat_ws(B, cap, W, H, dev)      # screen boxes + per-tile surfel lists (SDFR_PRIM_BINS workspace)
        # per-tile surfel lists (count -> scan -> fill, one workgroup per crop inside sdfr_surfels_forward) instead of every tile scanning
        # all boxes: same bits either way; the lists pay from a few crops per launch (B=64: splat forward 947 -> 551 us), at one crop the
        # distributed scan is the faster of the two (building the lists is a 13 us latency chain on one CU)
        self.binned = B >= 4
        if self.ragged:
            PS = self.PS
            self.color, self.mask, self.depth, self.nimg = f(B, 3, PS), f(B, 1, PS), f(B, 1, PS), f(B, 3, PS)
        else:
            self.color, self.mask, self.depth, self.nimg = f(B, 3, H, W), f(B, 1, H, W), f(B, 1, H, W), f(B, 3, H, W)
        self.aux = f(B, self.PS, 4)
        self.xyzf = f(B, cap, 3)
        # backward
        self.g_p, self.g_n, self.g_a = f(B, cap, 3), f(B, cap, 3), f(B, cap, 3)
        self.g_points, self.g_normals, self.g_pose = f(B, cap, 3), f(B, cap, 3), f(B, 16)
        self.g_latn = f(B, self.L)
        self.g_yaw, self.g_trans, self.g_latent = f(B), f(B, 3), f(B, self.L)
        self._graph = None
        # pose-only refinement (BASELINE configs[1] wording): with the latent fixed, sdf, band, Jacobian and surfels do not change between
        # iterations -- freeze_shape=True evaluates the decoder stages once (until the latent is set again) and every later forward() only
        # re-projects and splats.  Exact: the skipped kernels would reproduce the cached arrays bit for bit.
        self.freeze_shape = False
        self._shape_valid = False
        self.fused_tail = True      # one launch for the backward tail (False: the three separate kernels, same bits)
        self.fused_head = True      # one launch for surface projection + camera projection + screen boxes (False: three launches, same bits)

    @property
    def boxes(self):
        """the surfels' conservative screen boxes [B][cap][4] (x0, y0, x1, y1) at the head of the splat workspace"""
        return self.bbox[:self.B * self.cap * 4].view(self.B, self.cap, 4)

    # ------------------------------------------------------------------------------------------------------------------
    def set_extents(self, sizes_wh, K=None):
        """ragged mode: per-crop image sizes [(W_b, H_b)] * B and (optionally) intrinsics K (B,3,3) or (3,3); in place, so a captured graph
        stays valid.  Image regions beyond a crop's W_b H_b pixels are left as they are (never read)."""
        if not self.ragged:
            raise _lib.SdfrError("set_extents needs a BatchRenderer built with max_pixels")
        sizes = [(int(w), int(h)) for w, h in sizes_wh]
        if len(sizes) != self.B:
            raise _lib.SdfrError("set_extents: %d sizes for %d crops" % (len(sizes), self.B))
        for w, h in sizes:
            if w < 1 or h < 1 or w * h > self.PS or max(w, h) > self.max_side:
                raise _lib.SdfrError("crop of %dx%d pixels exceeds max_pixels %d / max_side %d" % (w, h, self.PS, self.max_side))
        self.sizes = sizes
        self.wh.copy_(torch.tensor(sizes, dtype=torch.int32))
        if K is not None:
            K = torch.as_tensor(K, dtype=torch.float32).cpu()
            if K.dim() == 2:
                K = K.unsqueeze(0).expand(self.B, 3, 3)
            self.K.copy_(K.contiguous())
            self.Kinv.copy_(torch.linalg.inv(K.float()).contiguous())            # primitives.py:204, on the host, once per crop set

    def image(self, b, name="color"):
        """(C, H_b, W_b) view of crop b's image `name` in 'color' | 'mask' | 'depth' | 'normals' (both layouts)"""
        t = {"color": self.color, "mask": self.mask, "depth": self.depth, "normals": self.nimg}[name]
        if not self.ragged:
            return t[b]
        w, h = self.sizes[b]
        return t[b, :, :w * h].view(t.shape[1], h, w)

    def set_params(self, yaw, trans, latent):
        self.yaw.copy_(yaw.reshape(self.B))
        self.trans.copy_(trans.reshape(self.B, 3))
        self.latent.copy_(latent.reshape(self.B, self.L))
        self._shape_valid = False
        self.clear_overflow()
        self.reset_guard()

    def invalidate_shape(self):
        """call after changing self.latent in place (freeze_shape mode): the next forward() re-evaluates decoder, band and Jacobian"""
        self._shape_valid = False
        if self.guarded:
            self.age.zero_()                     # the next step runs the half pass over the whole grid

    def clear_overflow(self):
        """forget the sticky truncation flags (new crops)"""
        self.over.zero_()

    def reset_guard(self):
        """float32_prefilter: new crops start with clean guard state -- the violation counters, the last deviation and the per-crop margin
        (back to the calibrated one) belong to the crops that were refined before, and a hard violation there must not make
        check_overflow() refuse every later, unrelated crop (refiners are cached and reused across Optimizer objects).  Called by
        set_params() and BatchRefiner.set_crops(); all in place, so a captured graph stays valid."""
        if self.guarded:
            self.age.zero_()                     # new crops: the next step runs the half pass over the whole grid
            self.violations.zero_()
            self.over.zero_()
            self.n_full.zero_()
            self.max_dev.zero_()
            self.margin_dev.fill_(self.margin)
            if self.audit:
                self.audit_dev.zero_()

    def forward(self, yaw=None, trans=None, latent=None, mlp_events=None, events=None):
        """mlp_events: optional (start, end) torch.cuda.Event pair recorded around the decoder-forward launch (bench.py roofline).
        events: optional dict of such pairs for other launches of the step: 'jacobian', 'splat_fwd' (and 'splat_bwd' in backward())."""
        with _lib.guard(self.dev):
            return self._forward(yaw, trans, latent, mlp_events, events or {})

    def backward(self, g_color=None, g_mask=None, g_depth=None, g_normals=None, g_xyzf=None, events=None):
        with _lib.guard(self.dev):
            return self._backward(g_color, g_mask, g_depth, g_normals, g_xyzf, events or {})

    def _forward(self, yaw, trans, latent, mlp_events, events):
        if yaw is not None:
            self.set_params(yaw, trans, latent)
        L = _lib.lib()
        P, st, ck = _lib.ptr, _lib.stream_ptr(), _lib.check
        B, G, cap, W, H = self.B, self.G, self.cap, self.W, self.H
        frozen = self.freeze_shape and self._shape_valid      # pose-only step: the decoder rows stay as they are, only pose / norm are rebuilt
        if self.creuse and self.fused and not frozen:
            ck(L.sdfr_params_plan(P(self.yaw), P(self.trans), P(self.latent), self.L, P(self.grid), G, B, P(self.inputs), P(self.pose), P(self.latnorm),
                                  self.lipschitz_plan, P(self.margin_dev), P(self.max_dev), P(self.lat_ref), P(self.age), self.max_reuse,
                                  P(self.reuse_flag), P(self.n_full), st), "sdfr_params_plan")
        else:
            ck(L.sdfr_params_forward(P(self.yaw), P(self.trans), P(self.latent), self.L, P(self.grid), G, B, None if frozen else P(self.inputs),
                                     P(self.pose), P(self.latnorm), st), "sdfr_params_forward")
        if mlp_events is not None:
            mlp_events[0].record()
        if self.freeze_shape and self._shape_valid:
            if mlp_events is not None:
                mlp_events[1].record()
        elif self.prefilter:
            if self.reuse:
                ck(L.sdfr_prefilter_plan(P(self.inputs), G, self.NI, self.L, B, self.lipschitz, P(self.margin_dev), P(self.max_dev), P(self.lat_ref),
                                         P(self.age), self.max_reuse, P(self.reuse_flag), P(self.n_full), st), "sdfr_prefilter_plan")
                ck(L.sdfr_mlp_forward_f16_skip(self.handle.h, P(self.inputs), B * G, P(self.sdf), P(self.reuse_flag), G, st),
                   "sdfr_mlp_forward_f16_skip")
                if self.fault is not None:
                    self.sdf.index_copy_(0, self.fault[0], self.fault[1])
                ck(L.sdfr_band_select_ex(P(self.sdf), G, B, self.thr, P(self.margin_dev), P(self.reuse_flag), P(self.cidx), cap, P(self.ccnt),
                                         P(self.cslot), P(self.scratch), P(self.over), 2, st), "sdfr_band_select_ex")
            else:
                ck(L.sdfr_mlp_forward_f16(self.handle.h, P(self.inputs), B * G, P(self.sdf), None, st), "sdfr_mlp_forward_f16")
                if self.fault is not None:
                    self.sdf.index_copy_(0, self.fault[0], self.fault[1])
                ck(L.sdfr_band_select_ex(P(self.sdf), G, B, self.thr, P(self.margin_dev), None, P(self.cidx), cap, P(self.ccnt), P(self.cslot),
                                         P(self.scratch), P(self.over), 2, st), "sdfr_band_select_ex")
            if self.audit:
                ck(L.sdfr_prefilter_audit_select(P(self.inputs), P(self.cslot), G, self.NI, B, self.audit_stride, P(self.audit_phase), P(self.audit_rows),
                                                 P(self.audit_src), P(self.audit_n), self.audit_cap, st), "sdfr_prefilter_audit_select")
                if self.audit_split:
                    ck(L.sdfr_mlp_forward_split_counted(self.handle.h, P(self.audit_rows), self.audit_cap, P(self.audit_n), P(self.audit_sdf), st),
                       "sdfr_mlp_forward_split_counted")
                else:
                    ck(L.sdfr_mlp_forward_counted(self.handle.h, P(self.audit_rows), self.audit_cap, P(self.audit_n), P(self.audit_sdf), 0, st),
                       "sdfr_mlp_forward_counted")
                ck(L.sdfr_prefilter_audit_check(P(self.sdf), P(self.audit_sdf), P(self.audit_src), P(self.audit_n), self.audit_cap, G, B, self.thr,
                                                P(self.reuse_flag) if self.reuse else None, P(self.audit_dev), P(self.violations), P(self.audit_phase), st),
                   "sdfr_prefilter_audit_check")
            # exact float32 sdf and Jacobian of the candidates (recomputing kernel, 16-row tiles), patched into the grid array
            ck(L.sdfr_mlp_jacobian(self.handle.h, P(self.inputs), G, B, P(self.cidx), cap, P(self.ccnt), P(self.Jc), P(self.sdf_band), None, None,
                                   0, st), "sdfr_mlp_jacobian")
            # exact values patched into the grid array + guard (deviation of the half pass at the candidates -> margin / violation counters)
            ck(L.sdfr_prefilter_guard2(P(self.sdf), P(self.sdf_band), P(self.cidx), G, B, cap, P(self.ccnt), P(self.margin_dev), P(self.max_dev),
                                       P(self.violations), P(self.reuse_flag) if self.reuse else None, st), "sdfr_prefilter_guard")
            ck(L.sdfr_band_select_ex(P(self.sdf), G, B, self.thr, None, None, P(self.idx), cap, P(self.cnt), None, P(self.scratch), P(self.over), 1, st),
               "sdfr_band_select_ex")
            ck(L.sdfr_gather_rows(P(self.J), P(self.Jc), self.NI, P(self.idx), P(self.cslot), G, B, cap, cap, P(self.cnt), st), "sdfr_gather_rows")
            if mlp_events is not None:
                mlp_events[1].record()
        elif self.creuse:
            cs = self.cstride
            if not self.fused:
                ck(L.sdfr_prefilter_plan(P(self.inputs), G, self.NI, self.L, B, self.lipschitz_plan, P(self.margin_dev), P(self.max_dev), P(self.lat_ref),
                                         P(self.age), self.max_reuse, P(self.reuse_flag), P(self.n_full), st), "sdfr_prefilter_plan")
            # full-grid pass of the crops whose candidate set is due (no masks: the Jacobian takes them from the candidate pass below)
            fwd_skip = L.sdfr_mlp_forward_f16_skip if (self.f16 or self.select_half) else L.sdfr_mlp_forward_skip
            ck(fwd_skip(self.handle.h, P(self.inputs), B * G, P(self.sdf), P(self.reuse_flag), G, st), "sdfr_mlp_forward_skip")
            if self.fault is not None:
                self.sdf.index_copy_(0, self.fault[0], self.fault[1])
            if self.fused:
                ck(L.sdfr_band_select_ex(P(self.sdf), G, B, self.thr, P(self.margin_dev), P(self.reuse_flag), P(self.cidx), cs, P(self.ccnt),
                                         P(self.cslot), P(self.scratch), P(self.over), 2, st), "sdfr_band_select_ex")
            else:
                ck(L.sdfr_band_select_skip(P(self.sdf), G, B, self.thr, P(self.margin_dev), P(self.reuse_flag), P(self.cidx), cs, P(self.ccnt),
                                           P(self.cslot), P(self.scratch), st), "sdfr_band_select_skip")
            if self.audit:
                # few crops per launch: every decoder pass of the step is ONE tile pass of latency with most CUs idle (25-50 tiles on 256 CUs), so
                # the audit's pass runs BESIDE the candidates' on a side stream (fork here, join at the end of forward(); capturable: the side
                # stream is forked from and joined into the capturing stream).  It reads rows OUTSIDE the candidates only; the main stream writes
                # candidate rows.  r06: up to 64 crops per launch -- a full chip gains too (the audit's workgroups fill the CUs that the pool launches
                # leave idle in their last, partial round: +1 % float16, +4 % exact float32 at 64 crops; tools/audit_side_ab.py).
                ast = st
                if self.audit_side:
                    self._side.wait_stream(torch.cuda.current_stream(self.dev))
                    ast = self._side.cuda_stream
                    self._side_pending = True
                ck(L.sdfr_prefilter_audit_select(P(self.inputs), P(self.cslot), G, self.NI, B, self.audit_stride, P(self.audit_phase), P(self.audit_rows),
                                                 P(self.audit_src), P(self.audit_n), self.audit_cap, ast), "sdfr_prefilter_audit_select")
                # float16: half | 2 = 128- / 64-row tiles of the same 32x32x16 products -- the bits of the full-grid launch.
                # float32: float32-GRADE values from the error-compensated split kernel (within 2.4e-7 of the exact kernel at 2.5x its speed: the
                # audit asks whether a row outside the candidates sits inside the band, against a proof that leaves it >= 0.3 margin outside);
                # decoder.candidate_audit_arith = "float32" takes the exact kernel
                if self.f16:
                    ck(L.sdfr_mlp_forward_counted(self.handle.h, P(self.audit_rows), self.audit_cap, P(self.audit_n), P(self.audit_sdf), 3, ast),
                       "sdfr_mlp_forward_counted")
                elif self.audit_split:
                    ck(L.sdfr_mlp_forward_split_counted(self.handle.h, P(self.audit_rows), self.audit_cap, P(self.audit_n), P(self.audit_sdf), ast),
                       "sdfr_mlp_forward_split_counted")
                else:
                    ck(L.sdfr_mlp_forward_counted(self.handle.h, P(self.audit_rows), self.audit_cap, P(self.audit_n), P(self.audit_sdf), 0, ast),
                       "sdfr_mlp_forward_counted")
                ck(L.sdfr_prefilter_audit_check(P(self.sdf), P(self.audit_sdf), P(self.audit_src), P(self.audit_n), self.audit_cap, G, B, self.thr,
                                                P(self.reuse_flag), P(self.audit_dev), P(self.violations), P(self.audit_phase), ast),
                   "sdfr_prefilter_audit_check")
            # every crop: the candidates through the same kernel (values + masks), written into the grid array
            # one or two crops per launch: tiles of half the size (twice the workgroups for the same rows; same bits per row)
            ht = 2 if self.quarter_tiles else (1 if self.half_tiles else 0)
            if self.fused:
                # r06: the candidate rows are read where they lie (no gathered copy), by a pool of workgroups over the live tiles; the band is
                # compacted straight from the candidate values (scatter + grid-wide selection + position map in one launch)
                ck(L.sdfr_mlp_forward_candidates(self.handle.h, P(self.inputs), G, B, P(self.cidx), cs, P(self.ccnt), P(self.csdf), P(self.cmask),
                                                 1 if self.f16 else 0, ht, st), "sdfr_mlp_forward_candidates")
                if mlp_events is not None:
                    mlp_events[1].record()
                ck(L.sdfr_candidate_band(P(self.sdf), P(self.csdf), P(self.cidx), G, B, cs, P(self.ccnt), self.thr, P(self.idx), cap, P(self.cnt),
                                         P(self.cpos), P(self.over), st), "sdfr_candidate_band")
            else:
                ck(L.sdfr_candidate_rows(P(self.inputs), G, self.NI, B, P(self.cidx), cs, P(self.ccnt), P(self.crow), st), "sdfr_candidate_rows")
                fwd_ragged = L.sdfr_mlp_forward_f16_ragged if self.f16 else L.sdfr_mlp_forward_ragged
                ck(fwd_ragged(self.handle.h, P(self.crow), B, cs, P(self.ccnt), P(self.csdf), P(self.cmask), ht, st), "sdfr_mlp_forward_ragged")
                ck(L.sdfr_scatter_values(P(self.sdf), P(self.csdf), P(self.cidx), G, B, cs, P(self.ccnt), st), "sdfr_scatter_values")
                if mlp_events is not None:
                    mlp_events[1].record()
                ck(L.sdfr_band_select(P(self.sdf), G, B, self.thr, P(self.idx), cap, P(self.cnt), None, P(self.scratch), st), "sdfr_band_select")
                ck(L.sdfr_candidate_band_map(P(self.idx), cap, P(self.cnt), P(self.cslot), G, B, cs, P(self.cpos), P(self.violations), st),
                   "sdfr_candidate_band_map")
            if "jacobian" in events:
                events["jacobian"][0].record()
            # (the mask-fed Jacobian reads no input rows: the gathered array is not needed)
            ck(L.sdfr_mlp_jacobian(self.handle.h, P(self.crow), cs, B, P(self.cpos), cap, P(self.cnt), P(self.J), P(self.sdf_band), P(self.csdf),
                                   P(self.cmask), (2 if self.f16 else 0) | (64 if self.quarter_tiles else (32 if self.half_tiles else 0)), st),
               "sdfr_mlp_jacobian")                                                   # [SDFR_JAC_HALF_TILES / SDFR_JAC_QUARTER_TILES]
            if "jacobian" in events:
                events["jacobian"][1].record()
        else:
            fwd = L.sdfr_mlp_forward_f16 if self.f16 else (L.sdfr_mlp_forward_split if self.split else L.sdfr_mlp_forward)
            ck(fwd(self.handle.h, P(self.inputs), B * G, P(self.sdf), P(self.mask_ws), st), "sdfr_mlp_forward")
            if mlp_events is not None:
                mlp_events[1].record()
            ck(L.sdfr_band_select_ex(P(self.sdf), G, B, self.thr, None, None, P(self.idx), cap, P(self.cnt), None, P(self.scratch), P(self.over), 1, st),
               "sdfr_band_select_ex")
            if "jacobian" in events:
                events["jacobian"][0].record()
            ck(L.sdfr_mlp_jacobian(self.handle.h, P(self.inputs), G, B, P(self.idx), cap, P(self.cnt), P(self.J), P(self.sdf_band), P(self.sdf),
                                   P(self.mask_ws), 2 if self.f16 else 0, st), "sdfr_mlp_jacobian")
            if "jacobian" in events:
                events["jacobian"][1].record()
        self._shape_valid = True
        xyz = self.inputs[:, self.NI - 3:]
        prim = 512 if self.binned else 0                                              # [SDFR_PRIM_BINS]
        if self.ragged:
            ck(L.sdfr_surfels_forward_r(P(xyz), self.NI, P(self.sdf), G, P(self.idx), P(self.J), self.NI, self.NI - 3, P(self.pose), P(self.K), B,
                                        cap, P(self.cnt), self.nocs_mode | 4 | (8 if self.binned else 0), P(self.wh), self.tiles_cap, _DIAM_DISC,
                                        P(self.points), P(self.normals), P(self.p_cam), P(self.n_cam), P(self.attr), P(self.fidx), P(self.fcnt),
                                        P(self.xyzf), P(self.fslot), P(self.bbox), st), "sdfr_surfels_forward_r")
            prim |= 256
        elif self.fused_head:
            # band rows -> surfels -> camera frame -> front-facing list -> screen boxes in one launch; nocs_mode | 4: the composited
            # attribute (col + 1) / 2 (rasterer.py:113-114) is written directly
            ck(L.sdfr_surfels_forward(P(xyz), self.NI, P(self.sdf), G, P(self.idx), P(self.J), self.NI, self.NI - 3, P(self.pose), P(self.K), B,
                                      cap, P(self.cnt), self.nocs_mode | 4 | (8 if self.binned else 0), W, H, _DIAM_DISC, P(self.points), P(self.normals), P(self.p_cam),
                                      P(self.n_cam), P(self.attr), P(self.fidx), P(self.fcnt), P(self.xyzf), P(self.fslot), P(self.bbox), st),
               "sdfr_surfels_forward")
            prim |= 256                                                               # SDFR_PRIM_BOXES_READY
        else:
            ck(L.sdfr_surface_project(P(xyz), self.NI, P(self.sdf), G, B, P(self.idx), cap, P(self.cnt), P(self.J), self.NI, self.NI - 3,
                                      P(self.points), P(self.nocs), P(self.normals), st), "sdfr_surface_project")
            # nocs_mode | 4: the projection writes the composited attribute (col + 1) / 2 (rasterer.py:113-114) and the front-facing xyzf rows
            ck(L.sdfr_project_dcm(P(self.pose), P(self.K), P(self.points), P(self.normals), None, B, cap, P(self.cnt), self.nocs_mode | 4, W, H,
                                  P(self.p_cam), P(self.n_cam), P(self.attr), None, P(self.fidx), P(self.fcnt), P(self.xyzf), P(self.fslot), st),
               "sdfr_project_dcm")
        if "splat_fwd" in events:
            events["splat_fwd"][0].record()
        if self.ragged:
            ck(L.sdfr_splat_forward_r(prim, P(self.K), P(self.Kinv), P(self.p_cam), P(self.n_cam), P(self.attr), B, cap, P(self.cnt), P(self.wh),
                                      self.PS, self.tiles_cap, _DIAM_DISC, _DEPTH_CONSTANT, P(self.bbox), P(self.color), P(self.mask), P(self.depth),
                                      P(self.nimg), P(self.aux), st), "sdfr_splat_forward_r")
        else:
            ck(L.sdfr_splat_forward(prim, P(self.K), P(self.Kinv), P(self.p_cam), P(self.n_cam), P(self.attr), None, None, None, None, B, cap, P(self.cnt), W, H, _DIAM_DISC,
                                    _DEPTH_CONSTANT, P(self.bbox), P(self.color), P(self.mask), P(self.depth), P(self.nimg), P(self.aux), st),
               "sdfr_splat_forward")
        if "splat_fwd" in events:
            events["splat_fwd"][1].record()
        if getattr(self, "_side_pending", False):               # the audit's side stream joins before anything later can touch its buffers
            torch.cuda.current_stream(self.dev).wait_stream(self._side)
            self._side_pending = False
        return {"color": self.color, "mask": self.mask, "depth": self.depth, "normals": self.nimg, "xyzf": self.xyzf, "nf": self.fcnt,
                "n": self.cnt}

    def _backward(self, g_color, g_mask, g_depth, g_normals, g_xyzf, events):
        L = _lib.lib()
        P, st, ck = _lib.ptr, _lib.stream_ptr(), _lib.check
        B, cap, W, H = self.B, self.cap, self.W, self.H

        def c(g, shape):
            if g is None:
                return None
            g = g.to(torch.float32).expand(shape).contiguous()
            return g

        g_color, g_mask = c(g_color, self.color.shape), c(g_mask, self.mask.shape)
        g_depth, g_normals = c(g_depth, self.depth.shape), c(g_normals, self.nimg.shape)
        if "splat_bwd" in events:
            events["splat_bwd"][0].record()
        if self.ragged:
            ck(L.sdfr_splat_backward_r(P(self.K), P(self.Kinv), P(self.p_cam), P(self.n_cam), P(self.attr), B, cap, P(self.cnt), P(self.wh), self.PS,
                                       _DIAM_DISC, _DEPTH_CONSTANT, P(self.aux), P(self.color), P(self.mask), P(self.depth), P(self.nimg),
                                       P(g_color), P(g_mask), P(g_depth), P(g_normals), P(self.g_p), P(self.g_n), P(self.g_a), st),
               "sdfr_splat_backward_r")
        else:
            ck(L.sdfr_splat_backward(0, P(self.K), P(self.Kinv), P(self.p_cam), P(self.n_cam), P(self.attr), None, None, None, None, B, cap, P(self.cnt), W, H, _DIAM_DISC,
                                     _DEPTH_CONSTANT, P(self.aux), P(self.color), P(self.mask), P(self.depth), P(self.nimg), P(g_color),
                                     P(g_mask), P(g_depth), P(g_normals), P(self.g_p), P(self.g_n), P(self.g_a), st), "sdfr_splat_backward")
        if "splat_bwd" in events:
            events["splat_bwd"][1].record()
        if g_xyzf is not None:
            g_xyzf = c(g_xyzf, self.xyzf.shape)
        if self.L <= 8 and self.fused_tail:
            # projection backward (with the (col + 1) / 2 map of the attribute and the gradient arriving through xyzf), latent gradient
            # and parameter gradients in one launch
            ck(L.sdfr_pose_latent_backward(P(self.pose), P(self.points), P(self.normals), P(self.g_p), P(self.g_n), P(self.g_a), B, cap,
                                           P(self.cnt), self.nocs_mode | 4, P(g_xyzf), P(self.fslot), None if self.freeze_shape else P(self.J), self.NI, self.L, P(self.yaw),
                                           P(self.latent), P(self.latnorm), None, P(self.g_pose), P(self.g_latn), P(self.g_yaw),
                                           P(self.g_trans), P(self.g_latent), st), "sdfr_pose_latent_backward")
            return self.g_yaw, self.g_trans, self.g_latent
        # the projection backward folds in the (col + 1) / 2 map of the attribute and the gradient arriving through xyzf
        ck(L.sdfr_project_dcm_bwd(P(self.pose), P(self.points), P(self.normals), P(self.g_p), P(self.g_n), P(self.g_a), B, cap,
                                  P(self.cnt), self.nocs_mode | 4, P(self.g_points), P(self.g_normals), None, P(self.g_pose), P(g_xyzf),
                                  P(self.fslot), st), "sdfr_project_dcm_bwd")
        ck(L.sdfr_surface_latent_grad(P(self.g_points), None, P(self.normals), P(self.J), self.NI, self.L, B, cap, P(self.cnt),
                                      P(self.g_latn), st), "sdfr_surface_latent_grad")
        ck(L.sdfr_params_backward(P(self.yaw), P(self.latent), self.L, P(self.latnorm), P(self.g_pose), P(self.g_latn), B, P(self.g_yaw),
                                  P(self.g_trans), P(self.g_latent), st), "sdfr_params_backward")
        return self.g_yaw, self.g_trans, self.g_latent

    def backward_solve(self, g_color, g_xyzf, kscale, sv):
        """r06, the refinement loop's backward in two launches: the splat backward for a colour gradient that arrives UN-normalised with the
        per-crop factor kscale[b, 0] (sdfr_losses_fused), then projection / latent / parameter gradients (g_xyzf times kscale[b, 1]) and the
        solver step of every crop (sv: the solver's buffers, BatchRefiner) in one launch.  Same bits as backward() + sdfr_solver_step."""
        with _lib.guard(self.dev):
            L = _lib.lib()
            P, st, ck = _lib.ptr, _lib.stream_ptr(), _lib.check
            B, cap = self.B, self.cap
            ck(L.sdfr_splat_backward_x(P(self.K), P(self.Kinv), P(self.p_cam), P(self.n_cam), P(self.attr), B, cap, P(self.cnt), self.W, self.H,
                                       P(self.wh) if self.ragged else None, self.PS, _DIAM_DISC, _DEPTH_CONSTANT, P(self.aux), P(self.color),
                                       P(g_color), P(kscale), P(self.g_p), P(self.g_n), P(self.g_a), P(self.bbox), st), "sdfr_splat_backward_x")
            ck(L.sdfr_pose_latent_solver(P(self.pose), P(self.points), P(self.normals), P(self.g_p), P(self.g_n), P(self.g_a), B, cap, P(self.cnt),
                                         self.nocs_mode | 4, P(g_xyzf), P(self.fslot), P(kscale), None if self.freeze_shape else P(self.J), self.NI,
                                         self.L, P(self.yaw), P(self.latent), P(self.latnorm), P(self.g_pose), P(self.g_latn), P(sv["params"]),
                                         P(sv["grads"]), P(sv["loss2d"]), P(sv["loss3d"]), P(sv["npairs"]), sv["w2"], sv["w3"], P(sv["adam_m"]),
                                         P(sv["adam_v"]), P(sv["adam_t"]), 0.01, 0.01, sv["lr_latent"], P(sv["total"]), P(sv["stepped"]), st),
               "sdfr_pose_latent_solver")

    # ------------------------------------------------------------------------------------------------------------------
    def overflow(self):
        """True if some crop's band (or candidate set) did not fit its capacity in ANY forward since the flags were last cleared (set_params /
        BatchRefiner.set_crops / a raising check_overflow) -- its surplus surfels were dropped.  The flags are sticky device words written by the
        selection kernels (r06: the counts themselves are overwritten by every forward, so a band that overflowed in iterations 5-40 of a
        graph-replayed refinement and fits again at the end used to pass).  Synchronises."""
        over = (self.over != 0).any() | (self.cnt > self.cap).any()
        if self.prefilter:
            over = over | (self.ccnt > self.cap).any()
        if self.creuse:
            over = over | (self.ccnt > self.cstride).any()
        return bool(over.item())

    def prefilter_report(self):
        """float32_prefilter only: {'violations': soft count, 'hard_violations': steps in which the half pass deviated by more than the
        margin at a candidate (a band row may have been missed), 'max_deviation': last step's, 'margin': current per-crop maximum}.
        One synchronisation."""
        if not self.guarded:
            # candidate reuse asked for but refused at construction (kernel error beyond the margin cap, no finite Lipschitz bound, unsupported
            # decoder): every step evaluates the whole grid; say so instead of returning nothing
            return None if self.reuse_off_reason is None else {"candidate_reuse": False, "reason": self.reuse_off_reason}
        v = self.violations.sum(0).tolist()
        rep = {"violations": int(v[0]), "hard_violations": int(v[1]), "max_deviation": float(self.max_dev.max()),
               "margin": float(self.margin_dev.max())}
        if self.creuse:
            rep.update({"candidate_reuse": True, "margin_grown_by_calibration": bool(self.margin_grown), "kernel_error_budget": self.select_error,
                        "half_kernel_deviation_sampled": self.f16_deviation_sampled, "e32": self.e32, "lipschitz_bound": self.lipschitz, "latent_share_of_margin": self.latent_share,
                        "full_grid_passes_per_crop": self.n_full.tolist()})
        if self.audit:
            rep["audit"] = {"stride": self.audit_stride, "rows_last_step": int(self.audit_n[0]), "steps": int(self.audit_phase[0]),
                            "reference_values": ("float16 (the mode's own kernel)" if self.f16 else
                                                 ("float32_split (error-compensated f16 MFMAs)" if self.audit_split else "float32 (the mode's own kernel)")) if self.creuse else
                            ("float32_split (error-compensated f16 MFMAs)" if self.audit_split else "float32"),
                            "max_deviation_at_non_candidates": float(self.audit_dev.max())}
        return rep

    def check_overflow(self):
        """Raise if the last forward dropped surfels (the reference has no capacity: a truncated shape must not pass silently).  One sync."""
        if self.overflow():
            worst = int(self.cnt.max()) if not self.guarded else max(int(self.cnt.max()), int(self.ccnt.max()))
            flags = self.over.tolist()
            self.over.zero_()                    # reported once: the renderer stays usable for the next crops
            raise _lib.SdfrError("a crop's band or candidate set exceeded the surfel capacity in some forward since the last check (sticky flags per "
                                 "crop %s; last counts up to %d) but BatchRenderer was built with cap=%d: rebuild it with a larger `cap` "
                                 "(default max(256, G/8))" % (flags, worst, self.cap))
        if self.creuse and int(self.violations[:, 1].sum()) > 0:
            hard = int(self.violations[:, 1].sum())
            self.violations[:, 1].zero_()
            raise _lib.SdfrError("candidate reuse: %d row(s) outside the candidate set were found inside the band (audit / band map): the "
                                 "band of those steps was incomplete; use decoder.candidate_reuse = False or a larger decoder.prefilter_margin" % hard)
        if self.prefilter and int(self.violations[:, 1].sum()) > 0:
            hard, worst_dev = int(self.violations[:, 1].sum()), float(self.max_dev.max())
            self.violations[:, 1].zero_()        # reported once: the renderer stays usable for the next crops (the grown margins remain)
            raise _lib.SdfrError("float32_prefilter: the half-operand pass deviated from the exact values by more than the safety margin "
                                 "(max deviation %g) in %d step(s): band rows may have been excluded; use precision=torch.float32 or a larger "
                                 "decoder.prefilter_margin" % (worst_dev, hard))

    def capture(self, grads_fn):
        """Capture forward -> grads_fn(outputs) -> backward in a HIP graph.  grads_fn maps the output dict to the keyword arguments of
        backward() using torch ops on static buffers only.  Returns a callable that replays the step (parameters are read from
        self.yaw / self.trans / self.latent, gradients land in self.g_yaw / self.g_trans / self.g_latent)."""
        s = torch.cuda.Stream(device=self.dev)
        s.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(s):
            for _ in range(2):
                self.backward(**grads_fn(self.forward()))
        torch.cuda.current_stream(self.dev).wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self.backward(**grads_fn(self.forward()))
        self._graph = g
        if not self.freeze_shape:
            return g.replay

        # pose-only mode: the warm-up above made the shape valid, so the captured launches are the pose-only ones.  After set_params() /
        # invalidate_shape() (a new latent) the decoder, band and Jacobian stages must run once before the graph is valid again: the
        # returned callable does that eager step itself instead of replaying stale surfels.
        def replay():
            if not self._shape_valid:
                self.backward(**grads_fn(self.forward()))
            else:
                g.replay()
        return replay
