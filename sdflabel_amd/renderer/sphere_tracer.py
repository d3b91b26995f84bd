"""SphereTracer -- per-ray sphere tracing of the DeepSDF level set (render mode of BASELINE.json's north_star wording).

NOT part of the reference: TRI-ML/sdflabel renders by splatting the surfels of a grid band (sdflabel_amd.Rasterer reproduces that to 1e-4).
This class is the other classic way to render an SDF, built on the same decoder kernels, offered beside the faithful path and labelled as a
different algorithm: there is no reference output it could be checked against.  Its oracle is oracle/sdf_oracle.py::sphere_trace /
sphere_trace_backward (numpy, the same step rule, polish and implicit-function gradient), and the splat renderer is a cross-check only
(silhouette / depth / NOCS agree up to the band thickness).

    forward(yaw[B], trans[B,3], latent[B,L]) -> {'color' (NOCS) [B,3,H,W], 'mask' [B,1,H,W], 'depth' [B,1,H,W], 'normals' [B,3,H,W]}

Everything is a HIP kernel behind the C ABI (csrc/trace.hip, csrc/mlp_kernel.h MODE 4) on pre-allocated buffers, with NO host synchronisation:
  sdfr_params_forward   pose [R(yaw) | t] with row 1 negated (optimizer.py:86-90), normalised latent (:96)
  sdfr_trace_setup      pixel rays in object space (o = -R^T t, d = R^T K^-1 [x, y, 1]) clipped against the cube [-1, 1]^3 -> active list
  sdfr_trace_march      while the device-side active count is >= tail_rows: decoder on the active rows (MFMA) + advance / retire / ballot
                        compaction per step; below it ONE launch of the decoder kernel in its looping mode marches the remaining rays to
                        termination (no per-step launch).  Speculation schedule `spec_levels` = [(first pass, samples per ray and pass), ...]
                        (default two levels: 4 samples from a pass index that depends on the crop size, 16 samples four passes later;
                        samples are accepted while each lies inside the previous one's safe sphere, spaced by the ray's radius ratio clamped
                        to [0.5, q_max]): every level is one launch with 64 / samples rays per tile, run by a pool of persistent workgroups;
                        the survivors of a level are re-packed for the next
  sdfr_trace_hits       hit pixels -> compact rows [latent, x0]
  hit pass              decoder value and input Jacobian at the hits (Newton polish, normals, d sdf / d latent): polish="decoder" in the
                        decoder's precision (float16: sdfr_mlp_forward_f16_counted with ReLU masks + the mask-fed half Jacobian),
                        polish="exact" always float32 (sdfr_mlp_jacobian, recomputing kernel)
  sdfr_trace_composite  one Newton step along non-grazing rays, then depth / NOCS colour / normals / mask images
  sdfr_trace_backward   image gradients -> pose and latent gradients through the implicit function f(o(θ) + λ d(θ), z(θ)) = 0 at the fixed
                        hit set (silhouette changes carry no gradient, as in the splat path), fixed-order sums; sdfr_params_backward
                        maps them to yaw / trans / latent.
"""
import numpy as np
import torch

from .. import _lib

_COUNTERS = 32                # include/sdfr.h SDFR_TRACE_COUNTERS


class _TraceFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tracer, yaw, trans, latent):
        out = tracer.render(yaw.detach(), trans.detach(), latent.detach())
        ctx.tracer = tracer
        # what the backward reads, copied: the tracer's buffers are overwritten by its next render (a few small device copies; the zero-copy
        # path is render() / backward())
        ctx.state = {k: getattr(tracer, k).clone() for k in ("pose", "hit_lam", "hit_slot", "J", "f0", "yaw", "latent", "latnorm")}
        # fresh tensors for autograd (the tracer's buffers are reused by the next call)
        color, mask, depth, normals = out["color"].clone(), out["mask"].clone(), out["depth"].clone(), out["normals"].clone()
        ctx.mark_non_differentiable(mask)
        return color, mask, depth, normals

    @staticmethod
    def backward(ctx, g_color, g_mask, g_depth, g_normals):
        g = ctx.tracer.backward(g_color, g_depth, g_normals, state=ctx.state)
        return None, g[0].clone(), g[1].clone(), g[2].clone()


def default_spec_from(rays_per_crop, half, cone=False):
    """first speculative pass of the default schedule.  Where the speculative passes pay depends on how many rays are still marching at that
    pass: K samples per ray cost K rows, and only once a pass is latency-bound (few tiles) are they free.  Measured optimum
    (tools/sphere_time.py --scan, one crop): without the cone phase 128x128 rays 8, 256x256 10, 512x512 13-14 (float16) / 18 (float32, whose
    64-row passes are matrix-bound); behind the cone phase (r04: 60 % of the pixel tiles never start a ray, the others start next to the
    surface) the float16 march pays earlier: 128x128 4 (3:7 1.26, 4:8 1.17, 5:9 1.20 ms), 256x256 6 (4:8 1.91, 5:9 1.84, 6:10 1.67, 10:13 1.71 ms),
    512x512 8 (8:11 3.64, 10:13 3.67, 13:16 3.85 ms) -- a function of the crop's ray count alone, fixed at construction, so the pass index
    still decides."""
    import math
    r = math.log2(max(int(rays_per_crop), 1) / 65536.0)
    if cone and half:
        return max(3, min(6 + round(r), 24))
    s = 10 + (round((1.5 if half else 4.0) * r) if r >= 0 else round(r))
    return max(4, min(s, 24))


def default_spec_levels(rays_per_crop, half, cone=False):
    """the default speculation schedule [(first pass index, samples per ray and pass), ...] of a crop with `rays_per_crop` pixels: a function of
    the crop's size only (never of the batch: a crop marches the same alone and inside a batch)"""
    sf = default_spec_from(rays_per_crop, half, cone)
    return [(sf, 4), (sf + (4 if (cone and half and rays_per_crop <= 65536) else 3), 16)]


def default_q_max():
    """upper clamp of the radius ratio q that spaces speculative samples (p_j = p_{j-1} + sigma q^j rho / |d|).  1.0 (r03) never guessed growing
    radii, so the rays that LEAVE the surface -- the grazing misses that end every march -- crept out at their smallest step; 1.5: 5 % fewer
    evaluations, 5-6 % less time at every size and batch (r04, profiles/r04_notes.md section 8); 2 and 3 measure the same (q rarely exceeds 1.5)"""
    return 1.5


class SphereTracer:
    def __init__(self, decoder, K, resolution_px, batch=1, steps=64, eps=2e-3, bound=1.0, near=1e-3, device="cuda", head_steps=None,
                 tail_rows=4096, spec_from=None, spec_k=None, sigma=0.9, spec_from2=None, spec_k2=None, polish=None,
                 cone_block=None, cone_steps=None, uniform_tiles=True, points=False, cone_spec_k=None, max_pixels=None, max_side=None,
                 spec_levels=None, q_max=None):
        """max_pixels / max_side (r04, ragged extents): every crop of the batch its OWN image size (W_b H_b <= max_pixels, sides <= max_side, default
        4 sqrt(max_pixels)) and intrinsics, set with set_extents(); all per-pixel arrays then hold slots of max_pixels pixels per crop ([B, C,
        max_pixels] images: image(b, name) gives the (C, H_b, W_b) view) and the kernels read the extents on the device, so one tracer (and one captured
        graph around it) serves any crop sizes within the caps.  The march schedule is then a function of the CAPACITY.  resolution_px: initial extents."""
        dev = torch.device(device)
        if dev.type != "cuda":
            raise _lib.SdfrError("SphereTracer runs on the GPU only")
        self.dev, self.B = dev, int(batch)
        self.W, self.H = int(resolution_px[0]), int(resolution_px[1])
        self.ragged = max_pixels is not None
        self.PS = int(max_pixels) if self.ragged else self.W * self.H                 # pixel slot per crop
        if self.ragged:
            import math
            self.max_side = int(max_side) if max_side is not None else min(self.PS, 4 * int(math.ceil(math.sqrt(self.PS))))
            if self.W * self.H > self.PS or max(self.W, self.H) > self.max_side:
                raise _lib.SdfrError("resolution_px %dx%d exceeds max_pixels %d / max_side %d" % (self.W, self.H, self.PS, self.max_side))
        self.steps, self.eps, self.bound, self.near = int(steps), float(eps), float(bound), float(near)
        # the device-side gate (count < tail_rows) is checked in each of the first head_steps steps; afterwards the tail takes whatever is left
        self.head_steps = min(self.steps, 24) if head_steps is None else int(head_steps)
        self.tail_rows = int(tail_rows)
        self.jac_rows32 = False         # True: exact-f32 value + Jacobian pass at the hits on 32-row tiles (measured equal to the 16-row tiles: 5.21 vs 5.14 ms)
        self.decoder = decoder
        self.handle = decoder.handle(dev)
        self.half = 1 if getattr(decoder, "mlp_precision", torch.float32) == torch.float16 else 0
        # speculative passes: from pass index spec_from on, spec_k samples per ray and pass in the looping kernel (accepted while each lies inside
        # the previous one's safe sphere).  Default 4 (with the second level below; one 256x256 crop, fwd+bwd: float16 3.76 -> 2.55 ms incl.
        # the half hit pass, float32 20.1 -> 18.5 ms); 1 = plain sphere tracing.
        self.spec_k = int(spec_k) if spec_k is not None else 4
        # LayerNorm decoders and hidden widths below 257 march with per-step launches of their own forward kernels: float32, plain tracing
        self.generic_march = bool(self.handle.has_ln or self.handle.hp != 512)
        if self.generic_march:
            if self.half:
                raise _lib.SdfrError("SphereTracer: the float16 march needs a 512-wide decoder without LayerNorm")
            self.spec_k, spec_k2 = 1, 1
        cone_on = (4 if cone_block is None else int(cone_block)) > 0
        spec_from_given = spec_from is not None
        if spec_from is None:
            spec_from = default_spec_from(self.PS, bool(self.half), cone_on)   # (per crop, not per batch: a crop renders the same alone or in a batch)
        self.spec_from, self.sigma = int(spec_from), float(sigma)
        if self.spec_k not in (1, 4):
            raise ValueError("spec_k must be 1 or 4")
        # second level: from pass index spec_from2 on (default spec_from + 3) the looping kernel's survivors -- the creeping rays that end the
        # march, scattered over the tiles -- are re-packed 64 / spec_k2 to a tile and take spec_k2 samples per pass (default 16 with spec_k 4)
        self.spec_k2 = int(spec_k2) if spec_k2 is not None else (16 if self.spec_k == 4 else 1)
        self.spec_from2 = int(spec_from2) if spec_from2 is not None else self.spec_from + (4 if (cone_on and self.half and not spec_from_given
                                                                                                      and self.PS <= 65536) else 3)
        if self.spec_k2 <= self.spec_k:
            self.spec_k2 = self.spec_k                                              # off
        elif self.spec_k2 not in (8, 16, 32, 64) or self.spec_from2 <= self.spec_from:
            raise ValueError("spec_k2 must be 8, 16, 32 or 64 (or <= spec_k: off), spec_from2 > spec_from")
        # The schedule the kernels get: levels [(first pass index, samples per ray and pass), ...], both ascending (include/sdfr.h sdfr_trace_march).
        # spec_levels given: that list.  Otherwise the two levels above -- or, with nothing about the schedule given at all, the default of
        # default_spec_levels(): four levels that spend the 64 rows of a tile on ever fewer rays (the march ends with a few hundred grazing
        # MISSES; r04 census in profiles/r04_notes.md section 8).  q_max: upper clamp of the radius ratio that spaces the speculative samples.
        legacy = spec_from_given or spec_k is not None or spec_k2 is not None or spec_from2 is not None
        if spec_levels is not None:
            self.levels = [(int(a), int(b)) for a, b in spec_levels]
        elif self.spec_k == 1:
            self.levels = []
        elif legacy:
            self.levels = [(self.spec_from, 4)] + ([(self.spec_from2, self.spec_k2)] if self.spec_k2 > self.spec_k else [])
        else:
            self.levels = default_spec_levels(self.PS, bool(self.half), cone_on)
        if self.generic_march:
            self.levels = []
        for i, (a, b) in enumerate(self.levels):
            if b not in (4, 8, 16, 32, 64) or a < 0 or (i and (a <= self.levels[i - 1][0] or b <= self.levels[i - 1][1])):
                raise ValueError("spec_levels: (first pass, samples) ascending in both, samples 4 / 8 / 16 / 32 / 64")
        if len(self.levels) > 6:
            raise ValueError("at most 6 speculation levels")
        self.q_max = float(q_max) if q_max is not None else (1.0 if (legacy or spec_levels is not None) else default_q_max())
        if self.levels:                                                             # (reported by bench.py: the first two levels)
            self.spec_k, self.spec_from = 4, self.levels[0][0]
            self.spec_from2, self.spec_k2 = (self.levels[1] if len(self.levels) > 1 else (self.spec_from, self.spec_k))
        else:
            self.spec_k = self.spec_k2 = 1
        self._levels_host = np.ascontiguousarray(np.asarray(self.levels, dtype=np.int32).reshape(-1, 2))      # (kept alive: the C call reads it)
        # the hit pass (decoder value + input Jacobian at the marched points: Newton polish, normals, implicit-function gradients): "exact" =
        # float32 whatever the decoder's precision; "decoder" = in the decoder's own precision -- with a float16 decoder the half forward with
        # ReLU masks + the mask-fed half Jacobian (what the splat path does at float16: 0.1 ms instead of 1.25 ms for 18 k hits; the surface
        # is then the HALF decoder's level set: depths within ~1e-3 of the exact polish).  Default: "decoder".
        self.polish = "decoder" if polish is None else str(polish)
        if self.polish not in ("exact", "decoder"):
            raise ValueError("polish must be 'exact' or 'decoder'")
        self.half_polish = bool(self.half) and self.polish == "decoder"
        # cone marching ahead of the per-ray march (default since r04: 4x4-pixel tiles; cone_block=0 turns it off): one ray per cone_block x
        # cone_block pixel tile until the SDF falls below the cone's radius; tiles whose cone leaves the cube are culled, the others' rays start
        # where their cone stopped (csrc/trace.hip sdfr_trace_cone; 3.2x fewer decoder evaluations on the bench crop)
        self.cone_block = 4 if cone_block is None else int(cone_block)
        # speculative cone passes (r04): cone_spec_k samples per cone and pass (accepted while inside the range the previous sample proved free).
        # A cone pass is one decoder pass of latency whatever its row count at one crop, so 4 samples x 4 passes (default) replace the 10
        # sequential plain passes -- the same culling for 2x the (few) cone evaluations; cone_spec_k=1, cone_steps=10: the plain cone march
        self.cone_spec_k = 4 if cone_spec_k is None else int(cone_spec_k)
        if self.cone_spec_k < 1 or self.cone_spec_k > 8:
            raise ValueError("cone_spec_k: 1 ... 8 samples per cone and pass")
        # passes: a crop of up to 4096 cones (256x256 at 4x4 pixels) is one round of 64-row tiles even with 4 samples each -- every pass costs one
        # decoder pass of latency, so 4 passes (1.70 ms per render against 2.03 with 10); a 512x512 crop's 16 k cones are matrix-bound, the extra
        # passes cull more and start the rays closer to the surface (3.64 against 4.2 ms).  A function of the crop size, not of the batch.
        bl = max(self.cone_block, 1)
        if self.ragged:                                                             # cone slots any admissible shape can need
            cones = (self.PS + bl * bl - 1) // (bl * bl) + (2 * self.max_side + bl - 1) // bl + 2
        else:
            cones = ((self.W + bl - 1) // bl) * ((self.H + bl - 1) // bl)
        self.cone_cap = cones
        nominal = (self.PS + bl * bl - 1) // (bl * bl) if self.ragged else cones         # (the schedule is a function of the pixel capacity)
        self.cone_steps = int(cone_steps) if cone_steps is not None else ((4 if nominal <= 4096 else 10) if self.cone_spec_k >= 4 else
                                                                          (6 if self.cone_spec_k >= 2 else 10))
        # uniform_tiles: the cone passes of a float16 decoder use ONE product shape whatever the device-side count (sdfr_mlp_forward_counted
        # half | 2), like the per-ray march (32x32x16 products in its 128- and 64-row head tiles and in the looping kernel with spec_k = 4):
        # a crop's rays then see the same decoder bits alone and inside a batch -> BatchRefiner(render="trace") refines a crop bit-identically
        # at any batch size.  False: 16-row tiles of 16x16x32 products for thin cone passes (a few us per pass faster at one crop, values
        # equal to float rounding only).
        self.uniform_tiles = bool(uniform_tiles)
        # ... and the hand-over from per-step launches to the looping kernel happens at a fixed PASS INDEX (spec_from) instead of when the
        # device-side TOTAL count drops below tail_rows: the two sides apply the same step rule but round a ray's state differently in the last
        # bit (separately compiled arithmetic), so a count-driven hand-over would make a crop's march depend on its batch mates
        self.march_tail_rows = 0 if (self.uniform_tiles and self.half and not self.generic_march and self.spec_k > 1) else self.tail_rows
        if self.cone_block and (self.cone_block < 2 or self.cone_steps < 1):
            raise ValueError("cone_block >= 2 (pixels), cone_steps >= 1")
        self.L = decoder.latent_size
        self.NI = self.L + 3
        K = torch.as_tensor(K, dtype=torch.float32)
        if K.dim() == 2:
            K = K.unsqueeze(0).expand(self.B, 3, 3)
        self.K = K.contiguous().to(dev)
        self.Kinv = torch.linalg.inv(K.cpu().float()).contiguous().to(dev)
        B, H, W = self.B, self.H, self.W
        n = B * self.PS
        f = lambda *s: torch.zeros(s, dtype=torch.float32, device=dev)
        i = lambda *s: torch.zeros(s, dtype=torch.int32, device=dev)
        self.yaw, self.trans, self.latent = f(B), f(B, 3), f(B, self.L)
        self.pose, self.latnorm, self.latn = f(B, 16), f(B), f(B, self.L)
        self.counters = i(_COUNTERS)
        self.pix, self.lam = [i(n), i(n), i(n)], [f(n, 4), f(n, 4), f(n, 4)]   # active lists (ping, pong, second tail stage): pixel, ray state (lam, rho, q, -)
        self.far, self.inputs, self.sdf = f(n), f(n, self.NI), f(n)
        self.tail_rows_buf = f(int(_lib.lib().sdfr_trace_pool()), 64 if self.levels else 16, self.NI)   # one tile of operand rows per pool workgroup
        if self.cone_block:
            nc = B * self.cone_cap
            self.cone = f(nc)                                                  # per pixel tile: start parameter or -1 (culled)
            self.cone_counters = i(_COUNTERS)
            self.cone_ids, self.cone_st, self.cone_aux = [i(nc), i(nc)], [f(nc, 4), f(nc, 4)], [f(nc, 2), f(nc, 2)]
            self.cone_inputs, self.cone_sdf = f(nc * self.cone_spec_k, self.NI), f(nc * self.cone_spec_k)
        self.hit_lam, self.hit_sdf, self.lam_s = f(n), f(n), f(n)
        self.hit_slot, self.idx = i(n), i(n)
        self.rows, self.J, self.f0 = f(n, self.NI), f(n, self.NI), f(n)
        if self.ragged:
            PS = self.PS
            self.color, self.mask, self.depth, self.normals = f(B, 3, PS), f(B, 1, PS), f(B, 1, PS), f(B, 3, PS)
            self.wh = torch.tensor([[W, H]] * B, dtype=torch.int32, device=dev)
            self.sizes = [(W, H)] * B
            self.tiles16_cap = (PS + 255) // 256 + (2 * self.max_side + 15) // 16 + 2          # (for sdfr_loss_2d_r on the traced image)
            self.ext = _lib.Extents(self.wh.data_ptr(), PS, self.cone_cap)
        else:
            self.color, self.mask, self.depth, self.normals = f(B, 3, H, W), f(B, 1, H, W), f(B, 1, H, W), f(B, 3, H, W)
        self.ws = f(int(_lib.lib().sdfr_trace_backward_ws_floats(B, self.PS, 1)))
        self.mask_ws = i(int(_lib.lib().sdfr_decoder_mask_words(self.handle.h, n))) if self.half_polish else None
        self.g_pose, self.g_latn = f(B, 16), f(B, self.L)
        self.g_yaw, self.g_trans, self.g_latent = f(B), f(B, 3), f(B, self.L)
        # points['xyzf'] of the refinement loop (optimizer.py:125): camera-frame hit points per crop in pixel order (sdfr_trace_points)
        self.ecap = self.PS
        if points:
            self.xyzf, self.ecnt, self.pt_slot = f(B, self.ecap, 3), i(B), i(n)
        else:
            self.xyzf = self.ecnt = self.pt_slot = None

    # ------------------------------------------------------------------------------------------------------------------
    def render(self, yaw=None, trans=None, latent=None, events=None):
        """forward without autograd: fills and returns the static image buffers.  yaw=None: the parameters already sit in self.yaw / .trans /
        .latent (BatchRefiner binds those to its flat parameter buffer).  events: optional {'march': (start, end)} torch.cuda.Event pairs
        recorded around the march (bench.py)."""
        L = _lib.lib()
        P, ck = _lib.ptr, _lib.check
        B, W, H = self.B, self.W, self.H
        n = B * W * H
        events = events or {}
        with _lib.guard(self.dev):
            st = _lib.stream_ptr()
            if yaw is not None:
                self.yaw.copy_(yaw.reshape(B)); self.trans.copy_(trans.reshape(B, 3)); self.latent.copy_(latent.reshape(B, self.L))
            ck(L.sdfr_params_forward(P(self.yaw), P(self.trans), P(self.latent), self.L, None, 1, B, None, P(self.pose), P(self.latnorm), st),
               "sdfr_params_forward")
            torch.div(self.latent, self.latnorm.unsqueeze(1), out=self.latn)                    # F.normalize (optimizer.py:96)
            if "march" in events:
                events["march"][0].record()
            if self.ragged:
                return self._render_ragged(L, P, ck, st, events)
            if self.cone_block:
                ck(L.sdfr_trace_cone(self.handle.h, P(self.pose), P(self.Kinv), P(self.latn), self.L, B, W, H, self.bound, self.near, self.eps,
                                     self.cone_block, self.cone_steps, self.cone_spec_k, self.sigma,
                                     (self.half | 2) if (self.half and self.uniform_tiles) else self.half,
                                     P(self.cone_counters), P(self.cone_ids[0]), P(self.cone_st[0]), P(self.cone_aux[0]),
                                     P(self.cone_ids[1]), P(self.cone_st[1]), P(self.cone_aux[1]), P(self.cone_inputs), P(self.cone_sdf), P(self.cone), st),
                   "sdfr_trace_cone")
            ck(L.sdfr_trace_setup2(P(self.pose), P(self.Kinv), P(self.latn), self.L, B, W, H, self.bound, self.near, P(self.counters), P(self.pix[0]),
                                   P(self.lam[0]), P(self.far), P(self.inputs), P(self.cone) if self.cone_block else None, self.cone_block,
                                   P(self.hit_lam), P(self.hit_sdf), st), "sdfr_trace_setup")
            ck(L.sdfr_trace_march(self.handle.h, P(self.pose), P(self.Kinv), P(self.latn), self.L, B, W, H, self.eps, self.steps,
                                  self.head_steps, self.march_tail_rows, self._levels_host.ctypes.data, len(self.levels), self.q_max, self.sigma,
                                  self.half, P(self.counters), P(self.pix[0]), P(self.lam[0]), P(self.pix[1]), P(self.lam[1]), P(self.pix[2]),
                                  P(self.lam[2]), P(self.far), P(self.inputs), P(self.sdf),
                                  P(self.tail_rows_buf), P(self.hit_lam), P(self.hit_sdf), st), "sdfr_trace_march")
            if "march" in events:
                events["march"][1].record()
            n_hits = self.counters[6:7]
            ck(L.sdfr_trace_hits(P(self.pose), P(self.Kinv), P(self.latn), self.L, B, W, H, P(self.hit_lam), P(n_hits), P(self.hit_slot),
                                 P(self.idx), P(self.rows), st), "sdfr_trace_hits")
            if self.half_polish:
                # half decoder value (+ ReLU masks) and mask-fed half Jacobian at the hits; both launches take the device-side count
                ck(L.sdfr_mlp_forward_f16_counted(self.handle.h, P(self.rows), n, P(n_hits), P(self.sdf), P(self.mask_ws), st),
                   "sdfr_mlp_forward_f16_counted")
                ck(L.sdfr_mlp_jacobian(self.handle.h, P(self.rows), n, 1, P(self.idx), n, P(n_hits), P(self.J), P(self.f0), P(self.sdf),
                                       P(self.mask_ws), 2 | 16, st), "sdfr_mlp_jacobian")     # [SDFR_JAC_MANY_ROWS]: 64-row tiles
            else:
                # exact-f32 decoder value and input Jacobian at the hits (recomputing kernel; rows beyond the device-side count are not touched)
                ck(L.sdfr_mlp_jacobian(self.handle.h, P(self.rows), n, 1, P(self.idx), n, P(n_hits), P(self.J), P(self.f0), None, None,
                                       16 if self.jac_rows32 else 0, st),      # [SDFR_JAC_MANY_ROWS]: thousands of hits -> 32-row tiles
                   "sdfr_mlp_jacobian")
            ck(L.sdfr_trace_composite(P(self.pose), P(self.Kinv), self.L, B, W, H, P(self.hit_lam), P(self.hit_slot), P(self.J), P(self.f0),
                                      P(self.color), P(self.mask), P(self.depth), P(self.normals), P(self.lam_s), st), "sdfr_trace_composite")
            if self.xyzf is not None:
                ck(L.sdfr_trace_points(P(self.Kinv), B, W, H, P(self.hit_slot), P(self.lam_s), P(self.xyzf), self.ecap, P(self.ecnt),
                                       P(self.pt_slot), st), "sdfr_trace_points")
        out = {"color": self.color, "mask": self.mask, "depth": self.depth, "normals": self.normals}
        if self.xyzf is not None:
            out["xyzf"], out["nf"] = self.xyzf, self.ecnt
        return out

    def _render_ragged(self, L, P, ck, st, events):
        """render() with per-crop extents: the same launches through the `_r` entry points (include/sdfr.h sdfr_extents)"""
        import ctypes
        B, n = self.B, self.B * self.PS
        E = ctypes.addressof(self.ext)
        if self.cone_block:
            ck(L.sdfr_trace_cone_r(self.handle.h, P(self.pose), P(self.Kinv), P(self.latn), self.L, B, E, self.bound, self.near, self.eps, self.cone_block,
                                   self.cone_steps, self.cone_spec_k, self.sigma, (self.half | 2) if (self.half and self.uniform_tiles) else self.half,
                                   P(self.cone_counters), P(self.cone_ids[0]), P(self.cone_st[0]), P(self.cone_aux[0]), P(self.cone_ids[1]),
                                   P(self.cone_st[1]), P(self.cone_aux[1]), P(self.cone_inputs), P(self.cone_sdf), P(self.cone), st), "sdfr_trace_cone_r")
        ck(L.sdfr_trace_setup_r(P(self.pose), P(self.Kinv), P(self.latn), self.L, B, E, self.bound, self.near, P(self.counters), P(self.pix[0]),
                                P(self.lam[0]), P(self.far), P(self.inputs), P(self.cone) if self.cone_block else None, self.cone_block, P(self.hit_lam),
                                P(self.hit_sdf), st), "sdfr_trace_setup_r")
        ck(L.sdfr_trace_march_r(self.handle.h, P(self.pose), P(self.Kinv), P(self.latn), self.L, B, E, self.eps, self.steps, self.head_steps,
                                self.march_tail_rows, self._levels_host.ctypes.data, len(self.levels), self.q_max, self.sigma, self.half, P(self.counters),
                                P(self.pix[0]), P(self.lam[0]), P(self.pix[1]), P(self.lam[1]), P(self.pix[2]), P(self.lam[2]), P(self.far), P(self.inputs),
                                P(self.sdf), P(self.tail_rows_buf), P(self.hit_lam), P(self.hit_sdf), st), "sdfr_trace_march_r")
        if "march" in events:
            events["march"][1].record()
        n_hits = self.counters[6:7]
        ck(L.sdfr_trace_hits_r(P(self.pose), P(self.Kinv), P(self.latn), self.L, B, E, P(self.hit_lam), P(n_hits), P(self.hit_slot), P(self.idx),
                               P(self.rows), st), "sdfr_trace_hits_r")
        if self.half_polish:
            ck(L.sdfr_mlp_forward_f16_counted(self.handle.h, P(self.rows), n, P(n_hits), P(self.sdf), P(self.mask_ws), st), "sdfr_mlp_forward_f16_counted")
            ck(L.sdfr_mlp_jacobian(self.handle.h, P(self.rows), n, 1, P(self.idx), n, P(n_hits), P(self.J), P(self.f0), P(self.sdf), P(self.mask_ws),
                                   2 | 16, st), "sdfr_mlp_jacobian")
        else:
            ck(L.sdfr_mlp_jacobian(self.handle.h, P(self.rows), n, 1, P(self.idx), n, P(n_hits), P(self.J), P(self.f0), None, None,
                                   16 if self.jac_rows32 else 0, st), "sdfr_mlp_jacobian")
        ck(L.sdfr_trace_composite_r(P(self.pose), P(self.Kinv), self.L, B, E, P(self.hit_lam), P(self.hit_slot), P(self.J), P(self.f0), P(self.color),
                                    P(self.mask), P(self.depth), P(self.normals), P(self.lam_s), st), "sdfr_trace_composite_r")
        if self.xyzf is not None:
            ck(L.sdfr_trace_points_r(P(self.Kinv), B, E, P(self.hit_slot), P(self.lam_s), P(self.xyzf), self.ecap, P(self.ecnt), P(self.pt_slot), st),
               "sdfr_trace_points_r")
        out = {"color": self.color, "mask": self.mask, "depth": self.depth, "normals": self.normals}
        if self.xyzf is not None:
            out["xyzf"], out["nf"] = self.xyzf, self.ecnt
        return out

    def set_extents(self, sizes_wh, K=None):
        """ragged mode: per-crop image sizes [(W_b, H_b)] * B and (optionally) intrinsics K (B,3,3) | (3,3); in place: a captured graph stays valid"""
        if not self.ragged:
            raise _lib.SdfrError("set_extents needs a SphereTracer built with max_pixels")
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
            self.Kinv.copy_(torch.linalg.inv(K.float()).contiguous())

    def image(self, b, name="color"):
        """(C, H_b, W_b) view of crop b's image `name` in 'color' | 'mask' | 'depth' | 'normals' (both layouts)"""
        t = {"color": self.color, "mask": self.mask, "depth": self.depth, "normals": self.normals}[name]
        if not self.ragged:
            return t[b]
        w, h = self.sizes[b]
        return t[b, :, :w * h].view(t.shape[1], h, w)

    def backward(self, g_color=None, g_depth=None, g_normals=None, state=None, g_xyzf=None, surfel=False):
        """gradients of the last render() w.r.t. yaw [B], trans [B,3], latent [B,L] (static buffers).  state: saved copies of the buffers of an
        earlier render (the autograd path).  g_xyzf [B, ecap, 3]: gradient w.r.t. the hit points (constructed with points=True).
        surfel=False: image-space derivative at the fixed pixels through the implicit function (a hit point moves along its pixel ray only);
        surfel=True: the hits are material points that move rigidly with the pose and along their normal with the latent, their NOCS colour is
        pose-independent -- the autograd semantics of the reference's surfels (grid.py:61, projection.py:53-58), the mode the refinement loop
        uses (include/sdfr.h sdfr_trace_refine_backward)."""
        L = _lib.lib()
        P, ck = _lib.ptr, _lib.check
        B, W, H = self.B, self.W, self.H
        S = state or {k: getattr(self, k) for k in ("pose", "hit_lam", "hit_slot", "J", "f0", "yaw", "latent", "latnorm")}

        def c(g, shape):
            return None if g is None else g.to(torch.float32).expand(shape).contiguous()

        g_color, g_depth, g_normals = c(g_color, self.color.shape), c(g_depth, self.depth.shape), c(g_normals, self.normals.shape)
        with _lib.guard(self.dev):
            st = _lib.stream_ptr()
            if g_xyzf is not None:
                if self.xyzf is None:
                    raise _lib.SdfrError("SphereTracer: g_xyzf needs a tracer built with points=True")
                g_xyzf = c(g_xyzf, self.xyzf.shape)
            if self.ragged:
                import ctypes
                ck(L.sdfr_trace_refine_backward_r(P(S["pose"]), P(self.Kinv), self.L, B, ctypes.addressof(self.ext), P(S["hit_lam"]), P(S["hit_slot"]),
                                                  P(S["J"]), P(S["f0"]), P(g_color), P(g_depth), P(g_normals), P(g_xyzf), P(self.pt_slot), self.ecap,
                                                  1 if surfel else 0, P(self.ws), P(self.g_pose), P(self.g_latn), st), "sdfr_trace_refine_backward_r")
            else:
                ck(L.sdfr_trace_refine_backward(P(S["pose"]), P(self.Kinv), self.L, B, W, H, P(S["hit_lam"]), P(S["hit_slot"]), P(S["J"]), P(S["f0"]),
                                                P(g_color), P(g_depth), P(g_normals), P(g_xyzf), P(self.pt_slot), self.ecap, 1 if surfel else 0,
                                                P(self.ws), P(self.g_pose), P(self.g_latn), st), "sdfr_trace_refine_backward")
            ck(L.sdfr_params_backward(P(S["yaw"]), P(S["latent"]), self.L, P(S["latnorm"]), P(self.g_pose), P(self.g_latn), B, P(self.g_yaw),
                                      P(self.g_trans), P(self.g_latent), st), "sdfr_params_backward")
        return self.g_yaw, self.g_trans, self.g_latent

    def forward(self, yaw, trans, latent):
        """differentiable render: images carry gradients to yaw, trans and latent (torch.autograd.Function over render() / backward())"""
        c, m, d, n = _TraceFn.apply(self, yaw, trans, latent)
        return {"color": c, "mask": m, "depth": d, "normals": n}

    __call__ = forward

    # ---- diagnostics (each reads device counters: one synchronisation) --------------------------------------------------------------
    def stats(self):
        c = self.counters.cpu().numpy()
        evals = int(c[4:6].view("uint64")[0])
        out = {"hits": int(c[6]), "unresolved": int(c[3]), "ray_evaluations": evals}
        if self.cone_block:
            cc = self.cone_counters.cpu().numpy()
            out["cone_evaluations"] = int(cc[4:6].view("uint64")[0])
            out["ray_evaluations"] += out["cone_evaluations"]                   # (decoder evaluations of the render: cones + rays)
            cone = self.cone.view(self.B, self.cone_cap)
            if self.ragged:
                # each crop's OWN tiles only: slots beyond ceil(W_b / block) * ceil(H_b / block) keep values of earlier extents
                cb = self.cone_block
                own = torch.tensor([((w + cb - 1) // cb) * ((h + cb - 1) // cb) for w, h in self.sizes], device=cone.device).view(-1, 1)
                out["culled_tiles"] = int(((cone < 0) & (torch.arange(self.cone_cap, device=cone.device).view(1, -1) < own)).sum())
            else:
                out["culled_tiles"] = int((cone < 0).sum())
        return out

    @property
    def n_hit(self):
        return int(self.counters[6])

    @property
    def n_unresolved(self):
        return int(self.counters[3])

    @property
    def hit_residual(self):
        """decoder value at the marched hit points (before the polish), in the hit pass's precision: exact float32 with polish="exact" or
        a float32 decoder, the half forward's value with polish="decoder" on a float16 decoder"""
        return self.f0[:self.n_hit]
