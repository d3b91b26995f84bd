/*
 * sdfr.h -- C ABI of libsdfr_hip.so: the MI355X (gfx950) differentiable SDF renderer hot path.
 *
 * The upstream reference (TRI-ML/sdflabel) exposes this path only as Python/PyTorch objects
 * (SURVEY.md 8b); there is no upstream FFI.  Each entry point below therefore names the reference
 * Python interface whose arithmetic it replaces (file:line relative to the reference root).  The
 * host-side mirror of those Python interfaces lives in sdflabel_amd/ and binds this library with
 * ctypes.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless the name starts with h_.
 *   - all floating-point data is float32, row-major, densely packed.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  All work is enqueued on it;
 *     no entry point synchronises except sdfr_decoder_create/destroy.
 *   - ragged per-crop data uses the layout [B][cap][...] with a device-side count per crop (`cnt`, int32[B]).
 *     cnt == NULL means "every crop holds exactly cap items".
 *   - return value: 0 on success, negative SDFR_E_* on error; sdfr_last_error() gives a message.
 */
#ifndef SDFR_H
#define SDFR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SDFR_OK 0
#define SDFR_E_INVALID (-1)   /* bad argument (shape, NULL pointer, unsupported architecture spec) */
#define SDFR_E_HIP (-2)       /* a HIP runtime call failed */
#define SDFR_E_UNSUPPORTED (-3)

#define SDFR_MAX_LAYERS 16
#define SDFR_TRACE_LEVELS 6     /* most speculation levels of a sphere-tracing march schedule (sdfr_trace_march) */
#define SDFR_TRACE_COUNTERS 32  /* int32 device counters of a march / a cone march (zeroed by sdfr_trace_setup / sdfr_trace_cone) */

#define SDFR_VERSION 400        /* what sdfr_version() of the library this header belongs to returns; a binding compares the two */

/* ABI version: bumped whenever an exported signature or a buffer size changes (300: the r04 argument lists of sdfr_trace_march /
 * sdfr_trace_cone and the 32-word SDFR_TRACE_COUNTERS; 400: the r06 fused entry points below -- sdfr_params_plan, sdfr_band_select_ex,
 * sdfr_mlp_forward_candidates, sdfr_candidate_band, sdfr_losses_fused, sdfr_splat_backward_x, sdfr_pose_latent_solver).  A caller built
 * against another header must refuse the library. */
int sdfr_version(void);
/* 0 for the product library.  Bit 0: built with SDFR_EXPERIMENT (kernel geometry / option A/B build of tools/ab_variant.sh);
 * bit 1: a timing-only ablation is compiled in and results are wrong by construction.  Bindings refuse a non-zero value. */
int sdfr_build_flags(void);
const char* sdfr_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * DeepSDF decoder  --  replaces Decoder.forward (sdfrenderer/deepsdf/networks/deep_sdf_decoder_scale.py:78-107)
 * and the autograd backward of it w.r.t. its input (the normals hook of sdfrenderer/grid.py:11-12,55-56 and
 * the latent gradient of pipelines/optimizer.py:156).
 *
 * The decoder is described by its EFFECTIVE linear layers (weight-norm g*v/||v|| already folded, as
 * nn.utils.weight_norm recomputes on every call, deep_sdf_decoder_scale.py:51-52):
 *   n_lin         number of nn.Linear layers (9 for the 8x512 DeepSDF decoder); the last one must have out_dim 1
 *   in_dim[l]     input width of layer l INCLUDING any re-injected input columns
 *   out_dim[l]    output width of layer l
 *   inj_n[l]      number of input columns concatenated to the activations BEFORE layer l
 *                 (L+3 when l is in latent_in, :90-91; 3 when xyz_in_all, :92-93; else 0); inj_n[0] must be 0
 *   inj_off[l]    first input column that is concatenated (0 for latent_in, L for xyz_in_all)
 *   h_W[l]        HOST pointer, float32 [out_dim[l]][in_dim[l]] row-major (torch nn.Linear.weight layout)
 *   h_b[l]        HOST pointer, float32 [out_dim[l]]
 *   n_inputs      width of an input row (L+3)
 *   use_tanh      apply tanh to the last linear before the final tanh (:96-97); the final tanh (:106-107) is always applied
 *   h_ln_w[l], h_ln_b[l]   HOST pointers to the nn.LayerNorm weight / bias [out_dim[l]] applied between layer l's linear and its ReLU
 *                 (the weight_norm=False decoder variant, :56-57,:99-101), or NULL; the arrays themselves may be NULL.  LayerNorm decoders
 *                 run in float32 without mask saving (their Jacobian recomputes the forward).
 * Hidden widths up to 512 are supported.
 */
typedef struct sdfr_decoder sdfr_decoder;

int sdfr_decoder_create(sdfr_decoder** out, int n_lin, const int* in_dim, const int* out_dim,
                        const int* inj_n, const int* inj_off, const float* const* h_W, const float* const* h_b,
                        const float* const* h_ln_w, const float* const* h_ln_b, int n_inputs, int use_tanh, int device);
int sdfr_decoder_destroy(sdfr_decoder* dec);
/* algorithmic multiply-accumulates per evaluated point (sum of in_dim*out_dim) */
int64_t sdfr_decoder_macs(const sdfr_decoder* dec);

/* sdf[i] = Decoder(inputs[i,:]) for i < n.   inputs [n][n_inputs], sdf [n].
 * Reference: pred_sdf_grid, _ = dsdf(inputs)  (pipelines/optimizer.py:99-101). */
int sdfr_mlp_forward(const sdfr_decoder* dec, const float* inputs, int64_t n, float* sdf,
                     uint32_t* mask_ws /* optional: sdfr_decoder_mask_words(dec, n) uint32 words; receives the ReLU masks
                                          (1 bit per hidden feature, point and layer) for a later sdfr_mlp_jacobian */,
                     void* stream);
/* the same with float16 operands on the matrix cores (weights and hidden activations rounded to half, float32 accumulation, bias,
 * ReLU and tanh) -- the decoder precision of the reference's default config (configs/config_refine.ini:19); inputs/outputs stay float32. */
int sdfr_mlp_forward_f16(const sdfr_decoder* dec, const float* inputs, int64_t n, float* sdf, uint32_t* mask_ws, void* stream);

/* sdfr_mlp_forward_f16 over the first *n_dev rows only (n_dev: device int32, clamped to n_max; no host read) -- the sphere tracer's hit pass
 * in the decoder's own half precision; mask_ws sized for n_max rows. */
int sdfr_mlp_forward_f16_counted(const sdfr_decoder* dec, const float* inputs, int64_t n_max, const int32_t* n_dev, float* sdf, uint32_t* mask_ws,
                                 void* stream);

/* the same with error-compensated float16 operands: each float32 weight and hidden activation x is carried as hi = half(x),
 * lo = half((x - hi) * 2^11) and every product as hi*hi + 2^-11 (hi*lo + lo*hi) on the f16 matrix cores, float32 accumulation
 * (~22 significand bits per product; the output differs from sdfr_mlp_forward by float32 summation-order noise, not by half
 * rounding).  Same replaced interface as sdfr_mlp_forward (deep_sdf_decoder_scale.py:78-107); mask_ws has the layout
 * sdfr_mlp_forward writes (pass mask_from_f16 = 0 to sdfr_mlp_jacobian).  Requires |hidden activation| < 65504. */
int sdfr_mlp_forward_split(const sdfr_decoder* dec, const float* inputs, int64_t n, float* sdf, uint32_t* mask_ws, void* stream);
/* ... over the first *n_dev rows (device int32, clamped to n_max; rows beyond are not touched): the audit rows of the two-stage evaluation */
int sdfr_mlp_forward_split_counted(const sdfr_decoder* dec, const float* inputs, int64_t n_max, const int32_t* n_dev, float* sdf, void* stream);
/* size (in uint32 words) of the mask workspace for n rows */
int64_t sdfr_decoder_mask_words(const sdfr_decoder* dec, int64_t n);

/* Input Jacobian of the decoder at selected rows:
 *   for crop b < B, slot s < cnt[b]:  r = row_base + b*rows_per_crop + idx[b*cap+s]
 *     J[b][s][:]      = d sdf(inputs[r,:]) / d inputs[r,:]      (n_inputs values)
 *     sdf_sel[b][s]   = sdf(inputs[r,:])                         (may be NULL)
 * Rows s < cnt[b] are fully written by the call (no prior initialisation needed); rows s >= cnt[b] are left untouched.
 * Reference: the xyz columns are what grid.py:55-56 captures through its hook for the band points; the latent
 * columns give d sdf/d latent, which autograd recomputes at optimizer.py:156. */
int sdfr_mlp_jacobian(const sdfr_decoder* dec, const float* inputs, int64_t rows_per_crop, int B,
                      const int32_t* idx, int cap, const int32_t* cnt, float* J, float* sdf_sel,
                      const float* sdf_full /* optional: output of the sdfr_mlp_forward call over the same rows */,
                      const uint32_t* mask_ws /* optional: masks that call saved; with both, no forward recomputation */,
                      int mask_from_f16 /* 0: mask_ws from sdfr_mlp_forward(_split); 1: from sdfr_mlp_forward_f16, float32 backward;
                                            2: from sdfr_mlp_forward_f16 and the backward also runs with half operands;
                                            | SDFR_JAC_MANY_ROWS: hint for the recomputing kernel (no masks) that the launch holds many
                                            thousands of rows (32-row tiles; results equal to float rounding, not bit for bit) */,
                      void* stream);

/* ------------------------------------------------------------------------------------------------
 * Zero-isosurface projection  --  replaces Grid3D.get_surface_points (sdfrenderer/grid.py:43-71)
 */

/* Order-preserving selection of the band |sdf| < thr (grid.py:64-66), per crop:
 *   idx[b][0..cnt[b])  = ascending local row indices g in [0,G) with |sdf[b*G+g]| < thr
 *   slot[b*G+g]        = position of g in idx[b] or -1          (may be NULL)
 * If a crop holds more than cap band points the surplus is dropped and cnt[b] is set to the TRUE count
 * (callers compare against cap).  scratch: int32[B * ceil(G/256)]. */
int sdfr_band_select(const float* sdf, int64_t G, int B, float thr, int32_t* idx, int cap, int32_t* cnt,
                     int32_t* slot, int32_t* scratch, void* stream);

/* points[b][s] = x - sdf * n_hat, n_hat = J_xyz/||J_xyz||, nocs = (points+1)/2   (grid.py:57-67)
 *   inputs [B*G][n_inputs] (xyz = last 3 columns), J [B][cap][n_inputs] from sdfr_mlp_jacobian, or
 *   (when Jstride == 3) raw normals d sum(sdf)/d xyz gathered by the caller.
 *   outputs points, nocs, normals: [B][cap][3]. */
int sdfr_surface_project(const float* xyz, int xyz_stride, const float* sdf, int64_t G, int B,
                         const int32_t* idx, int cap, const int32_t* cnt, const float* J, int Jstride, int Joff,
                         float* points, float* nocs, float* normals, void* stream);

/* Backward of the projection (n_hat is a constant, grid.py:56-58):
 *   g_sdf[b*G + idx] = -(g_points + g_nocs/2) . n_hat ; g_xyz[b*G+idx][:] = g_points + g_nocs/2   (others 0; buffers are
 *   fully overwritten).  g_nocs, g_xyz may be NULL. */
int sdfr_surface_project_bwd(const float* g_points, const float* g_nocs, const float* normals, int64_t G, int B,
                             const int32_t* idx, int cap, const int32_t* cnt, float* g_sdf, float* g_xyz, void* stream);

/* Two-stage band selection (optional): a half-operand decoder pass over the grid + sdfr_band_select with a safety margin gives
 * candidate rows; sdfr_mlp_jacobian without masks evaluates them exactly (float32 sdf + Jacobian).  sdfr_scatter_values writes the exact
 * values back, dst[b*G + idx[b][s]] = src[b][s] for s < cnt[b], so that the ordinary sdfr_band_select on dst yields the exact band of
 * grid.py:64; sdfr_gather_rows re-orders rows of ncol floats, out[b][e] = src[b][slot[b*G + idx[b][e]]] for e < cnt[b] (slot = the
 * grid-row -> candidate-slot map sdfr_band_select wrote; src has src_cap rows per crop). */
int sdfr_scatter_values(float* dst, const float* src, const int32_t* idx, int64_t G, int B, int cap, const int32_t* cnt, void* stream);
/* sdfr_band_select with a per-crop addition to the threshold (thr + thr_extra[b]): the candidate selection with a device-resident margin. */
int sdfr_band_select_margin(const float* sdf, int64_t G, int B, float thr, const float* thr_extra, int32_t* idx, int cap, int32_t* cnt,
                            int32_t* slot, int32_t* scratch, void* stream);
/* sdfr_scatter_values plus the run-time guard of the two-stage evaluation: max_dev[b] = max over the crop's candidates of
 * |sdf_grid (half pass) - sdf_exact|, measured while the exact values are patched in.  max_dev > margin[b]/2: violations[2b] += 1 and
 * margin[b] = max(margin[b], 4 max_dev) for the following selections; max_dev >= margin[b]: violations[2b+1] += 1 as well (a band row may
 * have been excluded in this step).  margin float[B] in/out, violations int32[B][2] in/out; no host synchronisation. */
int sdfr_prefilter_guard(float* sdf_grid, const float* sdf_exact, const int32_t* idx, int64_t G, int B, int cap, const int32_t* cnt,
                         float* margin, float* max_dev, int32_t* violations, void* stream);
/* Candidate-set reuse of the two-stage evaluation.  sdfr_prefilter_plan decides per crop, on the device, whether the candidates of the last
 * half pass still cover the band: reuse[b] = 1 while lip * |latent - latent of that pass| <= margin[b]/4, max_dev[b] <= margin[b]/2 and at
 * most max_reuse steps in a row (lip: Lipschitz constant of the decoder in the normalised latent; inputs: the decoder input rows, whose
 * first L columns hold it; lat_ref float[B][L], age int32[B]: state, zero-initialised).  The half pass, the candidate selection and the
 * guard then take the flags: flagged crops are skipped (sdfr_mlp_forward_f16_skip, sdfr_band_select_skip) / patched but not judged
 * (sdfr_prefilter_guard2).  n_full int32[B] (may be NULL; r05): += 1 for every crop that is NOT flagged, i.e. runs its full-grid pass. */
int sdfr_prefilter_plan(const float* inputs, int64_t G, int n_inputs, int L, int B, float lip, const float* margin, const float* max_dev,
                        float* lat_ref, int32_t* age, int max_reuse, int32_t* reuse, int32_t* n_full, void* stream);
int sdfr_mlp_forward_f16_skip(const sdfr_decoder* dec, const float* inputs, int64_t n, float* sdf, const int32_t* skip, int64_t rows_per_crop,
                              void* stream);
int sdfr_band_select_skip(const float* sdf, int64_t G, int B, float thr, const float* thr_extra, const int32_t* skip, int32_t* idx, int cap,
                          int32_t* cnt, int32_t* slot, int32_t* scratch, void* stream);
int sdfr_prefilter_guard2(float* sdf_grid, const float* sdf_exact, const int32_t* idx, int64_t G, int B, int cap, const int32_t* cnt,
                          float* margin, float* max_dev, int32_t* violations, const int32_t* reused, void* stream);
int sdfr_gather_rows(float* out, const float* src, int ncol, const int32_t* idx, const int32_t* slot, int64_t G, int B, int cap,
                     int src_cap, const int32_t* cnt, void* stream);
/* Audit of the two-stage evaluation (r04): the guard above observes the half pass only at the candidates; a row outside them that the half
 * pass misplaced by more than the margin is invisible to it.  Every step the NON-candidate rows of one slice of the grid -- the contiguous
 * rows [ph * ceil(G / stride), (ph + 1) * ceil(G / stride)), ph = *phase mod stride: every row once per `stride` steps -- are listed by
 * sdfr_prefilter_audit_select (rows
 * float[cap_rows][n_inputs] = their decoder input rows, src int32[cap_rows] = b * G + g, *n_audit = how many; cslot: the grid-row ->
 * candidate-slot map of the candidate selection), evaluated exactly by the caller (sdfr_mlp_forward_counted(dec, rows, cap_rows, n_audit,
 * sdf_exact, 0)) and judged by sdfr_prefilter_audit_check: |exact| < thr on such a row = a band row was excluded in this step ->
 * violations[2b+1] += 1 (a hard violation, like the guard's); audit_dev[b] = the largest |half - exact| seen on crops that ran the half pass
 * (reused[b] == 0; reused may be NULL); then *phase += 1.  phase, n_audit: device int32; no host synchronisation. */
int sdfr_prefilter_audit_select(const float* inputs, const int32_t* cslot, int64_t G, int n_inputs, int B, int stride, const int32_t* phase,
                                float* rows, int32_t* src, int32_t* n_audit, int cap_rows, void* stream);
int sdfr_prefilter_audit_check(const float* sdf_grid, const float* sdf_exact, const int32_t* src, const int32_t* n_audit, int cap_rows,
                               int64_t G, int B, float thr, const int32_t* reused, float* audit_dev, int32_t* violations, int32_t* phase,
                               void* stream);

/* Candidate reuse of the float16 decoder mode (r05; the reference's shipped precision, configs/config_refine.ini:19, re-evaluates all G grid
 * rows every iteration of pipelines/optimizer.py:96-104 although the latent moves by ~1e-6 per iteration).  The band is taken from the
 * candidate rows (|half sdf| < thr + margin at the crop's last full-grid pass) alone:
 *   sdfr_candidate_rows      rows[b][s][:] = inputs[b*G + cidx[b][s]][:] for s < ccnt[b] (ragged [B][stride] array, stride a multiple of 128;
 *                            the rows up to the next multiple of 128 are filled with a finite row)
 *   sdfr_mlp_forward_f16_ragged   the half decoder on crop b's first cnt[b] rows of such an array (whole 128-row tiles), masks saved: every
 *                            row gets the bits sdfr_mlp_forward_f16 gives it in a full-grid launch
 *   sdfr_scatter_values      writes them into the grid array; sdfr_band_select there returns the band (rows outside the candidates keep
 *                            their values of the last full pass, >= thr + margin in magnitude)
 *   sdfr_candidate_band_map  pos[b][e] = cslot[b*G + idx[b][e]]: the band rows' positions in the candidate array, for the mask-fed half
 *                            Jacobian (sdfr_mlp_jacobian(dec, rows, stride, B, pos, ...)); a band row that is no candidate counts a hard
 *                            violation (violations[2b+1], may be NULL)
 * sdfr_prefilter_plan (with a proven Lipschitz bound as `lip`) decides per crop when a full pass is due; the audit re-evaluates a rotating
 * slice of the other rows with the same kernel. */
int sdfr_candidate_rows(const float* inputs, int64_t G, int n_inputs, int B, const int32_t* cidx, int stride, const int32_t* ccnt, float* rows,
                        void* stream);
int sdfr_mlp_forward_f16_ragged(const sdfr_decoder* dec, const float* inputs, int B, int64_t rows_per_crop, const int32_t* cnt, float* sdf,
                                uint32_t* mask_ws, int half_tiles, void* stream);
int sdfr_candidate_band_map(const int32_t* idx, int cap, const int32_t* cnt, const int32_t* cslot, int64_t G, int B, int stride, int32_t* pos,
                            int32_t* violations, void* stream);
/* The same scheme with the EXACT float32 decoder (the parity path): sdfr_mlp_forward with per-crop skip flags (full-grid pass of the crops
 * whose candidate set is due; no masks) and over a ragged [B][rows_per_crop] array (rows_per_crop a multiple of 64; masks saved for the
 * float32 mask-fed Jacobian).  Every row gets the bits sdfr_mlp_forward gives it in a full-grid launch. */
int sdfr_mlp_forward_skip(const sdfr_decoder* dec, const float* inputs, int64_t n, float* sdf, const int32_t* skip, int64_t rows_per_crop,
                          void* stream);
int sdfr_mlp_forward_ragged(const sdfr_decoder* dec, const float* inputs, int B, int64_t rows_per_crop, const int32_t* cnt, float* sdf,
                            uint32_t* mask_ws, int half_tiles, void* stream);
/* half_tiles = 1 (float16 only; sdfr_mlp_forward_ragged returns SDFR_E_UNSUPPORTED): 64-row tiles instead of 128 -- the candidates of one or
 * two crops are a few dozen full-size tiles on 256 CUs, and a tile pass is pure latency; the same bits per row.  The mask-fed Jacobian then
 * needs SDFR_JAC_HALF_TILES in its mask_from_f16 argument. */

/* g_inputs[r][:] = g_sdf[r] * J[slot[r]][:]  for rows with slot[r] >= 0, else 0   (DeepSDF backward through the
 * cached band Jacobian).  n_uncached (device int32, may be NULL) receives the number of rows with g_sdf != 0 and
 * slot < 0, i.e. rows the cache does not cover. */
int sdfr_sdf_input_grad(const float* g_sdf, const int32_t* slot, const float* J, int n_inputs, int64_t G, int B, int cap,
                        float* g_inputs, int32_t* n_uncached, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Projection to the camera frame  --  replaces project_in_2D (sdfrenderer/renderer/projection.py:7-101), rot='dcm'
 *   pose [B][16] row-major 4x4 (only the top 3 rows are used, :34); K [B][9]
 *   in : points, normals, colors [B][cap][3]  (colors ignored when output_nocs != 0: 1 -> c = p*(-1,1,1), :53-55; 2 -> c = p, :147-149;
 *        5 / 6 = 1 / 2 with the compositing map (c+1)/2 of rasterer.py:113-114 already applied, so that col feeds sdfr_splat_* directly)
 *   out: p_cam, n_cam, col [B][cap][3]; uv [B][cap][2] (clamped, :88-93; may be NULL)
 *        front-facing filter n_cam.p_cam < 0 (:61-70): fidx [B][cap] ascending slots, fcnt [B]  (fidx/fcnt may be NULL);
 *        optional with fidx: xyzf [B][cap][3] = p_cam rows of the front-facing surfels in fidx order (points['xyzf'], rasterer.py:151)
 *        and fslot [B][cap] = position of each surfel in fidx or -1 (what the backward needs to route g_xyzf).
 */
int sdfr_project_dcm(const float* pose, const float* K, const float* points, const float* normals, const float* colors,
                     int B, int cap, const int32_t* cnt, int output_nocs, int res_x, int res_y,
                     float* p_cam, float* n_cam, float* col, float* uv, int32_t* fidx, int32_t* fcnt, float* xyzf, int32_t* fslot,
                     void* stream);

/* Batched path, one launch: sdfr_surface_project (band rows -> points, unit normals; grid.py:57-67) + sdfr_project_dcm in a NOCS colour
 * mode (projection.py:34-70) + the conservative disc screen boxes and their 8x8-pixel tile lists that sdfr_splat_forward would otherwise
 * build (bbox = its workspace; pass primitive | SDFR_PRIM_BOXES_READY to it; bbox may be NULL).  output_nocs | 8: bbox is the large
 * workspace of sdfr_splat_ws_words(B, cap, res_x, res_y) words and the tile lists are built too (then also pass SDFR_PRIM_BINS to
 * sdfr_splat_forward; building the lists costs one workgroup per crop ~13 us, which pays from about four crops per launch); without it
 * bbox is int32[B][cap][4].  Same arithmetic and outputs as the separate calls. */
int sdfr_surfels_forward(const float* xyz, int xyz_stride, const float* sdf, int64_t G, const int32_t* idx, const float* J, int Jstride,
                         int Joff, const float* pose, const float* K, int B, int cap, const int32_t* cnt, int output_nocs, int res_x,
                         int res_y, float diam, float* points, float* normals, float* p_cam, float* n_cam, float* col, int32_t* fidx,
                         int32_t* fcnt, float* xyzf, int32_t* fslot, int32_t* bbox, void* stream);

/* Backward: g_points, g_normals, g_colors [B][cap][3] (g_colors NULL when output_nocs), g_pose [B][16].  g_xyzf [B][cap][3] (may be
 * NULL): gradient w.r.t. the xyzf rows, added to g_p_cam through fslot. */
int sdfr_project_dcm_bwd(const float* pose, const float* points, const float* normals,
                         const float* g_p_cam, const float* g_n_cam, const float* g_col,
                         int B, int cap, const int32_t* cnt, int output_nocs,
                         float* g_points, float* g_normals, float* g_colors, float* g_pose, const float* g_xyzf, const int32_t* fslot,
                         void* stream);

/* ------------------------------------------------------------------------------------------------
 * Surfel splat + depth-softmax composite  --  replaces the primitives of sdfrenderer/renderer/primitives.py and the compositing of
 * Rasterer.forward (sdfrenderer/renderer/rasterer.py:92-144) without ever forming the N x P tensors.
 *   primitive 0 'disc'        inside_surfel(diam=0.04, softclamp=False, depth_constant=150)   primitives.py:165-242  (the optimizer's)
 *   primitive 1 'circle'      inside_circle(diam=0.02, softclamp=True, depth_constant=100)     primitives.py:4-71
 *   primitive 2 'circle_opt'  inside_circle_opt(diam=0.025, depth_constant=10000)              primitives.py:74-162
 *   Kinv [B][9] = inverse(K.float()) (primitives.py:204), K [B][9]
 *   p_cam, n_cam, attr [B][cap][3]   attr is the composited colour attribute AFTER the (c+1)/2 mapping
 *   uv [B][cap][2], znorm [B]        primitives 1,2 only: clamped pixel projections (sdfr_project_dcm) and ||depths||_2 per crop
 *   bg [B][3][H][W], bg_logit [B]    optional background row (add_bg): its colour and its logit (min over the surfels' logits - 1,
 *                                    primitives.py:65,147,235); both NULL for bg=None
 *   images: color [B][3][H][W], mask [B][H][W], depth [B][H][W], normals [B][3][H][W] (any may be NULL)
 *   aux [B][H*W][4]: per-pixel state for the backward (nu, max logit, softmax denominator, clamp gates)
 */
#define SDFR_JAC_MANY_ROWS 16
/* Since SDFR_VERSION 400 the mask workspace has ONE layout whatever launch saved it (64 bytes per row and layer, rows in blocks of 128:
 * csrc/mlp_kernel.h sdfr_mask_dword), so the two flags below no longer select anything; they are accepted and ignored. */
#define SDFR_JAC_HALF_TILES 32  /* (until version 300) the masks were saved by a forward on half-size tiles (sdfr_mlp_forward_ragged / _f16_ragged with half_tiles = 1) */
#define SDFR_JAC_QUARTER_TILES 64  /* (until version 300) ... on quarter-size tiles (sdfr_mlp_forward_candidates with half_tiles = 2: 32-row tiles, float16) */
#define SDFR_PRIM_BOXES_READY 256   /* OR into `primitive` of sdfr_splat_forward: bbox_ws already holds the surfels' screen boxes and tile lists */
#define SDFR_PRIM_BINS 512          /* OR into `primitive`: bbox_ws is the LARGE workspace of sdfr_splat_ws_words() and per-tile surfel lists are
                                       built in it and used; without the flag bbox_ws only needs int32[B][cap][4] (boxes) and every 8x8 tile
                                       scans all boxes -- same bits either way */
/* words of the large splat workspace: the conservative screen boxes int32[B][cap][4] followed, per crop, by the 8x8-pixel tile lists built
 * from them (count -> scan -> fill): tile_off[T+2] | tile_list[32*cap], T = ceil(W/8)*ceil(H/8) */
int64_t sdfr_splat_ws_words(int B, int cap, int W, int H);
int sdfr_splat_forward(int primitive, const float* K, const float* Kinv, const float* p_cam, const float* n_cam, const float* attr,
                       const float* uv, const float* znorm, const float* bg, const float* bg_logit,
                       int B, int cap, const int32_t* cnt, int W, int H, float diam, float depth_constant,
                       int32_t* bbox_ws /* workspace: int32[B][cap][4], or sdfr_splat_ws_words(B, cap, W, H) words with SDFR_PRIM_BINS */,
                       float* color, float* mask, float* depth, float* normals, float* aux, void* stream);

/* Backward w.r.t. p_cam, n_cam, attr given the gradients of the four images (any may be NULL; a non-NULL gradient
 * needs the corresponding forward image, which carries the softmax-backward sum <grad, output>).  The background logit is a
 * constant here. */
int sdfr_splat_backward(int primitive, const float* K, const float* Kinv, const float* p_cam, const float* n_cam, const float* attr,
                        const float* uv, const float* znorm, const float* bg, const float* bg_logit,
                        int B, int cap, const int32_t* cnt, int W, int H, float diam, float depth_constant,
                        const float* aux, const float* color, const float* mask, const float* depth, const float* normals,
                        const float* g_color, const float* g_mask, const float* g_depth, const float* g_normals,
                        float* g_p_cam, float* g_n_cam, float* g_attr, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Ragged extents (r04): every crop of a batch its OWN image size and intrinsics, as the crops of the reference pipeline have
 * (utils/refinement.py:586-609 adjust_intrinsics_crop: area-normalised, aspect kept; pipelines/refine_css.py:117-129,203-223 builds a
 * Rasterer per crop).  The `_r` entry points take the extents as DATA -- wh int32[B][2] = (W_b, H_b) on the device, next to the per-crop
 * K / Kinv [B][9] -- instead of launch constants, so one captured launch sequence serves any crop sizes within the caps:
 *   images / aux / gradients live in slots of pix_stride pixels per channel: color [B][3][pix_stride], mask / depth [B][pix_stride],
 *   aux [B][pix_stride][4]; crop b uses the first W_b H_b entries of each channel, rows of W_b pixels;
 *   tiles_cap >= ceil(W_b / 8) ceil(H_b / 8) for every crop (launch bound and tile-list layout of the splat workspace,
 *   sdfr_splat_ws_words_r words); tiles16_cap likewise for the 16 x 16 tiles of the 2-D loss.
 * Arithmetic per crop exactly as the fixed-extent calls: a crop's results are bit-identical to rendering it alone at its own size.
 * Disc primitive (the optimizer's), no background.
 * CALLER'S CONTRACT (the extents live on the device, the entry points cannot check them): for every crop  1 <= W_b, 1 <= H_b,
 * W_b * H_b <= pix_stride,  ceil(W_b/8) * ceil(H_b/8) <= tiles_cap,  ceil(W_b/16) * ceil(H_b/16) <= tiles16_cap,  and for the tracer
 * ceil(W_b/block) * ceil(H_b/block) <= cone_cap.  The Python layer validates them in set_extents() (sdflabel_amd/batch.py,
 * renderer/sphere_tracer.py) before they reach the device.  Defence in depth (r05): the kernels treat a crop with W_b < 1, H_b < 1,
 * W_b * H_b > pix_stride or more 8x8 tiles than tiles_cap as EMPTY (nothing rendered, zero loss and gradients, its slot untouched) instead
 * of writing out of bounds (tests/test_gpu_ragged.py); cone_cap and tiles16_cap remain the caller's to guarantee. */
int64_t sdfr_splat_ws_words_r(int B, int cap, int tiles_cap);
int sdfr_surfels_forward_r(const float* xyz, int xyz_stride, const float* sdf, int64_t G, const int32_t* idx, const float* J, int Jstride,
                           int Joff, const float* pose, const float* K, int B, int cap, const int32_t* cnt, int output_nocs,
                           const int32_t* wh, int tiles_cap, float diam, float* points, float* normals, float* p_cam, float* n_cam, float* col,
                           int32_t* fidx, int32_t* fcnt, float* xyzf, int32_t* fslot, int32_t* bbox, void* stream);
int sdfr_splat_forward_r(int flags /* SDFR_PRIM_BOXES_READY | SDFR_PRIM_BINS */, const float* K, const float* Kinv, const float* p_cam,
                         const float* n_cam, const float* attr, int B, int cap, const int32_t* cnt, const int32_t* wh, int pix_stride,
                         int tiles_cap, float diam, float depth_constant, int32_t* bbox_ws, float* color, float* mask, float* depth,
                         float* normals, float* aux, void* stream);
int sdfr_splat_backward_r(const float* K, const float* Kinv, const float* p_cam, const float* n_cam, const float* attr, int B, int cap,
                          const int32_t* cnt, const int32_t* wh, int pix_stride, float diam, float depth_constant, const float* aux,
                          const float* color, const float* mask, const float* depth, const float* normals, const float* g_color,
                          const float* g_mask, const float* g_depth, const float* g_normals, float* g_p_cam, float* g_n_cam, float* g_attr,
                          void* stream);
/* sdfr_loss_2d on ragged extents; scratch float[3 * B * tiles16_cap] */
int sdfr_loss_2d_r(const float* rend, const float* target, int B, const int32_t* wh, int pix_stride, int tiles16_cap, float diam,
                   float threshold_nocs, float weight, float* loss, float* g_rend, int32_t* nvalid, float* scratch, void* stream);

/* The dense weight matrix the reference's standalone primitives return (prob_color[:, 0, :] of inside_surfel / inside_circle /
 * inside_circle_opt, primitives.py:71,162,243): weights [B][rows][W*H], rows = cap (+1 background row when bg_logit != NULL), from the
 * per-pixel state `aux` of a sdfr_splat_forward call over the same surfels (with the same bg_logit).  `weights` must be zero-filled; only
 * covered (surfel, pixel) pairs are written.  Not used by the renderer itself. */
int sdfr_splat_weights(int primitive, const float* K, const float* Kinv, const float* p_cam, const float* n_cam, const float* uv,
                       const float* znorm, const float* bg_logit, int B, int cap, const int32_t* cnt, int W, int H, float diam,
                       float depth_constant, const float* aux, float* weights, void* stream);
/* Its backward w.r.t. p_cam / n_cam given g_weights [B][rows][W*H] and wsum[b][pix] = sum_j weights_j * g_weights_j (the softmax-backward
 * sum over ALL rows of the pixel, background row included).  Coverage masks, the per-pixel norm and the background logit are constants, as
 * in the reference's autograd (primitives.py:55,59,226,228). */
int sdfr_splat_weights_backward(int primitive, const float* K, const float* Kinv, const float* p_cam, const float* n_cam, const float* uv,
                                const float* znorm, int has_bg_row, int B, int cap, const int32_t* cnt, int W, int H, float diam,
                                float depth_constant, const float* aux, const float* g_weights, const float* wsum, float* g_p_cam,
                                float* g_n_cam, void* stream);

/* The same three calls with the primitive's clamp configuration spelled out (r05) -- what the reference's standalone functions accept
 * beyond the calls Rasterer.forward makes (rasterer.py:92-104):
 *   clamp_alt = 0   the renderer's clamp: disc hard (softclamp=False), circle / circle_opt sigmoid (softclamp=True)
 *   clamp_alt = 1   the other one: disc softclamp=True -- inside_surfel's OWN default, primitives.py:174,217-218: the mask
 *                   sigmoid((diam - d) * c) > 0 holds until exp overflows, so practically every surfel covers every pixel (dense work,
 *                   whole-image screen boxes); circle / circle_opt softclamp=False -- a hard edge (:51-53,:120)
 *   clamp_constant  softclamp_constant of the sigmoid clamps (> 0; 3 / 5 / 5 are the functions' defaults, :13,:83,:175); ignored by hard clamps
 * sdfr_splat_forward / _weights / _weights_backward are these with (0, default constant).  clamp_alt = 1 computes its own screen boxes
 * (no SDFR_PRIM_BOXES_READY / SDFR_PRIM_BINS). */
int sdfr_splat_forward_clamp(int primitive, const float* K, const float* Kinv, const float* p_cam, const float* n_cam, const float* attr,
                             const float* uv, const float* znorm, const float* bg, const float* bg_logit, int B, int cap, const int32_t* cnt,
                             int W, int H, float diam, float depth_constant, int clamp_alt, float clamp_constant, int32_t* bbox_ws,
                             float* color, float* mask, float* depth, float* normals, float* aux, void* stream);
int sdfr_splat_weights_clamp(int primitive, const float* K, const float* Kinv, const float* p_cam, const float* n_cam, const float* uv,
                             const float* znorm, const float* bg_logit, int B, int cap, const int32_t* cnt, int W, int H, float diam,
                             float depth_constant, int clamp_alt, float clamp_constant, const float* aux, float* weights, void* stream);
int sdfr_splat_weights_backward_clamp(int primitive, const float* K, const float* Kinv, const float* p_cam, const float* n_cam, const float* uv,
                                      const float* znorm, int has_bg_row, int B, int cap, const int32_t* cnt, int W, int H, float diam,
                                      float depth_constant, int clamp_alt, float clamp_constant, const float* aux, const float* g_weights,
                                      const float* wsum, float* g_p_cam, float* g_n_cam, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Batched refinement step glue  --  the per-iteration host tensor algebra of pipelines/optimizer.py:86-100, for B crops
 * at once and entirely on the device (no host synchronisation anywhere in a step).
 */

/* pose[b] = [R_y(yaw_b) | t_b] with the rotation's row 1 negated (optimizer.py:87-90, utils/refinement.py:108-125);
 * inputs[b*G+g] = [latent_b / max(||latent_b||, 1e-12), grid[g]]  (optimizer.py:96-100);  latnorm[b] = the norm used.
 * inputs == NULL: pose and latnorm only (pose-only refinement of a frozen shape; grid may then be NULL too). */
int sdfr_params_forward(const float* yaw, const float* trans, const float* latent, int L, const float* grid, int64_t G, int B,
                        float* inputs, float* pose, float* latnorm, void* stream);

/* g_latn[b][:] = sum_s -( (g_points + g_nocs/2)_s . n_hat_s ) * J[b][s][0:L]   -- backward of grid.py:61,67 chained with the
 * decoder's input gradient summed over the expanded latent rows (what autograd does at optimizer.py:156).  g_nocs may be NULL. */
int sdfr_surface_latent_grad(const float* g_points, const float* g_nocs, const float* normals, const float* J, int n_inputs, int L,
                             int B, int cap, const int32_t* cnt, float* g_latn, void* stream);

/* g_yaw[B], g_trans[B][3] from g_pose[B][16]; g_latent[B][L] from g_latn through F.normalize. */
int sdfr_params_backward(const float* yaw, const float* latent, int L, const float* latnorm, const float* g_pose, const float* g_latn,
                         int B, float* g_yaw, float* g_trans, float* g_latent, void* stream);

/* The backward tail of the batched step in one launch (latent sizes 1..8): sdfr_project_dcm_bwd (with its optional g_xyzf / fslot and
 * colour-map handling), sdfr_surface_latent_grad (g_latn[b][c] = sum_s -(g_points_s . normals_s) J[b][s][c]) and sdfr_params_backward,
 * with the same fixed-order reductions -- the results are bit-identical to calling the three.  g_points may be NULL (not stored).
 * J == NULL: pose gradients only (the latent is not a variable: pose-only refinement); g_latn and g_latent are zero-filled.
 * Replaces the autograd of pipelines/optimizer.py:86-100 + sdfrenderer/renderer/projection.py:34-70 + sdfrenderer/grid.py:61. */
int sdfr_pose_latent_backward(const float* pose, const float* points, const float* normals, const float* g_p_cam, const float* g_n_cam,
                              const float* g_col, int B, int cap, const int32_t* cnt, int output_nocs, const float* g_xyzf,
                              const int32_t* fslot, const float* J, int n_inputs, int L, const float* yaw, const float* latent,
                              const float* latnorm, float* g_points, float* g_pose, float* g_latn, float* g_yaw, float* g_trans,
                              float* g_latent, void* stream);

/* padded front-facing selection (points['xyzf'], rasterer.py:151): out[b][j] = src[b][idx[b][j]] for j < cnt[b], 0 beyond;
 * and its backward dst[b][idx[b][j]] += src[b][j]. */
int sdfr_gather_rows3(float* out, const float* src, const int32_t* idx, int B, int cap, const int32_t* cnt, void* stream);
int sdfr_scatter_add_rows3(float* dst, const float* src, const int32_t* idx, int B, int cap, const int32_t* cnt, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Losses and solver step of the refinement loop (pipelines/optimizer.py:144-157, 166-237), B crops at once, device resident.
 */

/* compute_loss_3d (optimizer.py:166-198): exact nearest lidar point (lidar/scale, :84) of every estimated point est[b][j], j < ecnt[b];
 * pairs with distance < threshold/scale[b]; loss[b] = mean pair distance (0 without pairs).  g_est [B][ecap][3] and g_scale [B] receive
 * weight * d loss / d(est, scale).  npairs[b] = number of pairs, or -1 when either cloud is empty (the loop skips the crop, :127-129).
 * scratch: 3 * B * ceil(ecap / 64) floats (partial sums of the two-pass reduction). */
int sdfr_loss_3d(const float* est, const int32_t* ecnt, int ecap, const float* lidar, const int32_t* lcnt, int lcap, const float* scale,
                 float threshold, float weight, int B, float* loss, float* g_est, float* g_scale, int32_t* npairs, float* scratch,
                 void* stream);

/* compute_loss_2d (optimizer.py:200-237) on rend, target [B][3][H][W]: per rendered non-zero pixel the smallest distance to the target
 * weighted by clamp(diam - pixel distance, 0); loss[b] = mean of the minima below threshold_nocs (NaN if none, 0 without rendered pixels,
 * exactly as the reference); g_rend = weight * d loss / d rend; nvalid[b] = number of pixels in the mean. */
int sdfr_loss_2d(const float* rend, const float* target, int B, int H, int W, float diam, float threshold_nocs, float weight,
                 float* loss, float* g_rend, int32_t* nvalid, float* scratch /* float[3 * B * ceil(W/16) * ceil(H/16)] */, void* stream);

/* MultipleOptimizer.step (optimizer.py:13-23,34-52): Adam(lr_adam, betas .9/.999, eps 1e-8) on yaw and trans, SGD on scale (lr_scale) and
 * latent (lr_latent).  params / grads are ONE flat structure-of-arrays buffer [ yaw(B) | trans(B,3) | scale(B) | latent(B,L) ].
 * total[b] = w3*loss3d + w2*loss2d; a crop is skipped (stepped[b] = 0, no state change) when npairs[b] < 0, total is NaN or total == 0
 * (optimizer.py:127-129,149-151). */
int sdfr_solver_step(float* params, const float* grads, int L, const float* loss2d, const float* loss3d, const int32_t* npairs,
                     float w2, float w3, float* adam_m, float* adam_v, int32_t* adam_t, float lr_adam, float lr_scale, float lr_latent,
                     int B, float* total, int32_t* stepped, void* stream);

/* ------------------------------------------------------------------------------------------------
 * r06: the refinement iteration in 12 launches instead of 21 (the per-annotation call of pipelines/refine_css.py:203-223 is a chain of
 * latency-bound launches: every one removed is ~5 us of a 0.3 ms iteration).  Each entry point below is the fusion of entry points above and
 * returns THEIR bits (GPU tests compare the two launch sequences array by array).
 */
/* sdfr_params_forward + sdfr_prefilter_plan (optimizer.py:86-100 + the candidate-reuse plan). */
int sdfr_params_plan(const float* yaw, const float* trans, const float* latent, int L, const float* grid, int64_t G, int B, float* inputs,
                     float* pose, float* latnorm, float lip, const float* margin, const float* max_dev, float* lat_ref, int32_t* age,
                     int max_reuse, int32_t* reuse, int32_t* n_full, void* stream);
/* sdfr_band_select_skip with a STICKY truncation flag: over[b] |= over_bit whenever crop b's selection holds more than cap rows (cnt[b] is
 * overwritten by every selection; the reference has no capacity, grid.py:64-66, so a truncated band must never pass silently).  thr_extra,
 * skip, slot and over may be NULL.  Launches of up to 8 crops with a flag array take one workgroup per crop (one launch instead of two). */
int sdfr_band_select_ex(const float* sdf, int64_t G, int B, float thr, const float* thr_extra, const int32_t* skip, int32_t* idx, int cap,
                        int32_t* cnt, int32_t* slot, int32_t* scratch, int32_t* over, int over_bit, void* stream);
/* sdfr_candidate_rows + sdfr_mlp_forward(_f16)_ragged without the gathered copy: row s < cnt[b] of crop b is
 * inputs[b * rows_per_crop_in + cidx[b][s]]; values -> sdf [B][stride], masks -> mask_ws.  half: the float16 kernel (half_tiles: 64-row
 * tiles); else the exact-float32 kernel.  stride a multiple of the tile (128 / 64).  (deep_sdf_decoder_scale.py:78-107 on the candidate rows) */
int sdfr_mlp_forward_candidates(const sdfr_decoder* d, const float* inputs, int64_t rows_per_crop_in, int B, const int32_t* cidx, int64_t stride,
                                const int32_t* cnt, float* sdf, uint32_t* mask_ws, int half, int half_tiles, void* stream);
/* sdfr_scatter_values + sdfr_band_select + sdfr_candidate_band_map: the band of a crop whose candidate set is valid lies inside the candidates,
 * which are listed in ascending grid-row order -- idx[b][e] = grid row of the e-th candidate with |csdf| < thr, pos[b][e] = its position in the
 * candidate array, cnt[b] their number; the candidate values are also written into sdf_grid.  over (may be NULL): sticky flags, |= 1 band > cap,
 * |= 2 candidates > stride.  (grid.py:64-66 on the candidate rows) */
int sdfr_candidate_band(float* sdf_grid, const float* csdf, const int32_t* cidx, int64_t G, int B, int stride, const int32_t* ccnt, float thr,
                        int32_t* idx, int cap, int32_t* cnt, int32_t* pos, int32_t* over, void* stream);
/* sdfr_loss_2d(_r) + sdfr_loss_3d in one launch (optimizer.py:166-237).  wh == NULL: dense H x W images; else ragged extents (pix_stride,
 * tiles16_cap as sdfr_loss_2d_r).  g_rend / g_est receive the UN-normalised gradients and kscale float[B][2] the per-crop factors
 * (weight / count, or 0) the consumers multiply on load: sdfr_splat_backward_x (kscale[2b]) and sdfr_pose_latent_solver (kscale[2b+1]).
 * Two launches (pixel + pairs passes together; one tiny finalize pass) where sdfr_loss_2d + sdfr_loss_3d take four. */
int sdfr_losses_fused(const float* rend, const float* target, int B, int H, int W, const int32_t* wh, int pix_stride, int tiles16_cap, float diam,
                      float threshold_nocs, float weight2d, float* loss2d, float* g_rend, int32_t* nvalid, float* scratch2, const float* est,
                      const int32_t* ecnt, int ecap, const float* lidar, const int32_t* lcnt, int lcap, const float* scale, float threshold3d,
                      float weight3d, float* loss3d, float* g_est, float* g_scale, int32_t* npairs, float* scratch3, float* kscale,
                      void* stream);
/* sdfr_splat_backward(_r) of the disc primitive for a colour gradient alone, scaled by kscale[2 b] on load (primitives.py:209-242 backward).
 * bbox_ws (may be NULL): the splat workspace this step's sdfr_surfels_forward(_r) / sdfr_splat_forward(_r) filled -- the surfels' screen boxes at
 * its head are then read instead of recomputed (same boxes, same results). */
int sdfr_splat_backward_x(const float* K, const float* Kinv, const float* p_cam, const float* n_cam, const float* attr, int B, int cap,
                          const int32_t* cnt, int W, int H, const int32_t* wh, int pix_stride, float diam, float depth_constant, const float* aux,
                          const float* color, const float* g_color, const float* kscale, float* g_p_cam, float* g_n_cam, float* g_attr,
                          const int32_t* bbox_ws, void* stream);
/* sdfr_pose_latent_backward (g_xyzf scaled by kscale[2 b + 1] on load) + sdfr_solver_step.  g_yaw / g_trans / g_latent are the sections of
 * `grads`, whose scale section the loss launch has written (optimizer.py:156 backward, :13-23 step). */
int sdfr_pose_latent_solver(const float* pose, const float* points, const float* normals, const float* g_p_cam, const float* g_n_cam,
                            const float* g_col, int B, int cap, const int32_t* cnt, int output_nocs, const float* g_xyzf, const int32_t* fslot,
                            const float* kscale, const float* J, int n_inputs, int L, const float* yaw, const float* latent, const float* latnorm,
                            float* g_pose, float* g_latn, float* params, float* grads, const float* loss2d, const float* loss3d,
                            const int32_t* npairs, float w2, float w3, float* adam_m, float* adam_v, int32_t* adam_t, float lr_adam,
                            float lr_scale, float lr_latent, float* total, int32_t* stepped, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Sphere-tracing render mode  --  NOT in the reference (its renderer splats surfels of a grid band); the mode BASELINE.json's north_star words
 * literally: per-ray march with a decoder evaluation per step and ballot compaction of terminated rays.  No parity claim.
 *   rays in object space: p_cam = R p + t (pose [B][16]), pixel ray K^-1 [x, y, 1] (Kinv [B][9]); lam = camera-frame depth along the ray.
 *   state: counters int32[3] (rotating active-ray counts, on the device), pix int32[n_max] (crop*W*H + pixel), lam float[n_max], two of each
 *   (ping-pong), far float[B*W*H], inputs float[n_max][L+3] decoder input rows of the active rays (latn [B][L] = normalised latent),
 *   hit_lam / hit_sdf float[B*W*H] (zero-filled by the caller; lam and decoder value of the rays that reached |sdf| < eps).
 */
/* decoder forward over the first *n_dev rows (device int32, clamped to n_max); half != 0: half operands on the matrix cores.
 * float32, 512-wide decoders: two launches per call, 64-row and 16-row tiles; the one that does not fit the count exits at once
 * (a thin step is one decoder pass of latency per workgroup: 0.12 ms on 16-row tiles, 0.44 ms on 64-row tiles).
 * half | 2 (half operands only): one product shape whatever the count -- 128-row tiles while they fill the chip, 64-row tiles of the same
 * 32x32x16 products below -- so that a row's value has the same bits in every launch (a crop then marches identically alone and in a batch;
 * the 16-row tiles of plain `half` use 16x16x32 products, whose summation order differs). */
int sdfr_mlp_forward_counted(const sdfr_decoder* dec, const float* inputs, int64_t n_max, const int32_t* n_dev, float* sdf, int half,
                             void* stream);
/* all pixels of all crops: slab test against the cube [-bound, bound]^3; hits enter the active list (counters[0]) at lam = max(entry, near).
 * cone (optional, from sdfr_trace_cone with the same cone_block): per pixel block the parameter its rays start from instead, or -1: the
 * block's rays are misses. */
int sdfr_trace_setup(const float* pose, const float* Kinv, const float* latn, int L, int B, int W, int H, float bound, float near,
                     int32_t* counters, int32_t* pix,
                     float* lam /* ray state float[n][4]: lam = next sample, rho = |sdf| of the previous sample, q = ratio of the last two radii, - */,
                     float* far, float* inputs, const float* cone, int cone_block, void* stream);
/* ... which also zero-fills hit_lam / hit_sdf (float[B*W*H], both or neither NULL): no separate fills before a march */
int sdfr_trace_setup2(const float* pose, const float* Kinv, const float* latn, int L, int B, int W, int H, float bound, float near,
                      int32_t* counters, int32_t* pix, float* lam, float* far, float* inputs, const float* cone, int cone_block, float* hit_lam,
                      float* hit_sdf, void* stream);
/* Cone marching ahead of the per-ray march (optional): ONE ray through the centre of every block x block pixel tile stands for its pixels --
 * all pixel rays share origin and parametrisation, so the tile's rays at parameter lam lie within lam * delta of the centre ray's point
 * (delta = max |d_corner - d_centre|).  v = decoder(centre point): v - lam delta > eps -> nothing within the cone's cross-section, advance by
 * (v - lam delta) / (|d_c| + delta); <= eps -> the tile's rays start their own march there; past the cube's far side for all of them ->
 * culled; after cone_steps passes the cones stop where they are.  cone float[B][ceil(W/block) * ceil(H/block)]: start parameter or -1.
 * A centre point outside the cube is evaluated clamped into it, v = sqrt(clamp distance^2 + max(f, 0)^2): no extrapolated decoder value is trusted.
 * spec_k (1 ... 8; r04): speculative cone passes -- a pass evaluates spec_k samples of a cone, p_0 = lam, p_j = p_{j-1} + sigma q^j a_prev
 * (a_prev = the cone's previous advance, q = ratio of its last two advances in [0.5, 1.5]); sample j counts only inside the range its
 * predecessor proved free, so the accepted prefix is a valid, shorter-stepped cone march (4 samples x 4 passes cull what 10 plain passes cull).
 * counters: device int32[SDFR_TRACE_COUNTERS] (zeroed here; [0..2] rotating ROW counts, [4..5] one uint64 = decoder evaluations); ids0/st0/aux0, ids1/st1/aux1:
 * ping-pong cone lists (int32[n], float[n][4], float[n][2], n = B * tiles); inputs float[n * spec_k][L+3], sdf float[n * spec_k] scratch.
 * half | 2: see sdfr_mlp_forward_counted.  No host synchronisation. */
int sdfr_trace_cone(const sdfr_decoder* dec, const float* pose, const float* Kinv, const float* latn, int L, int B, int W, int H, float bound,
                    float near, float eps, int block, int cone_steps, int spec_k, float sigma, int half, int32_t* counters, int32_t* ids0,
                    float* st0, float* aux0, int32_t* ids1, float* st1, float* aux1, float* inputs, float* sdf, float* cone, void* stream);
/* march step `step` (0, 1, ...): sdf = decoder values of the active rays (counters[step % 3] of them); lam += sdf / |d|; rays with
 * |sdf| < eps are recorded in hit_lam / hit_sdf and retired, rays past `far` are retired, the rest are compacted into pix_out / lam_out /
 * inputs (count in counters[(step + 1) % 3]) */
int sdfr_trace_step(const float* pose, const float* Kinv, const float* latn, int L, int W, int H, float eps, const float* sdf,
                    int32_t* counters, int step, int64_t n_max, const int32_t* pix_in, const float* lam_in, int32_t* pix_out, float* lam_out,
                    const float* far, float* inputs, float* hit_lam, float* hit_sdf, void* stream);

/* The whole march in ONE call, no host synchronisation (counters: device int32[SDFR_TRACE_COUNTERS], zeroed by sdfr_trace_setup -- [0..2]
 * rotating active counts, [3] rays unresolved when the step budget ran out, [4..5] one uint64 = decoder evaluations of the march, [6] hits,
 * [7 + i] rays handed from stage i of the looping kernel to stage i + 1, [16 + i] tile counter of stage i).  While the device-side count is
 * >= tail_rows a step is the decoder on the active rows + the step kernel; below it ONE launch of the decoder kernel in its looping mode takes
 * the remaining rays on (ray state in registers, no per-step launch, no compaction); the gate is evaluated on the device in each of the first
 * head_steps steps, then an unconditional launch takes what is left (tail_rows = 0: the hand-over happens at pass index head_steps, whatever
 * the count -- a crop then marches the same alone and inside a batch).
 * levels: HOST array int32[n_levels][2] = (first pass index, samples per ray and pass), both ascending, samples 4 / 8 / 16 / 32 / 64,
 * n_levels <= SDFR_TRACE_LEVELS: from pass index levels[i][0] on a pass evaluates levels[i][1] samples per ray (p_0 = lam, p_j = p_{j-1} +
 * sigma q^j rho / |d|; q = ratio of the ray's last two radii clamped to [0.5, q_max], q_max >= 1) and accepts the prefix in which every sample
 * lies inside the previous one's safe sphere -- a valid sphere-tracing sequence, nothing skipped; the pass index alone decides (head_steps is
 * clamped to levels[0][0]), so a ray's samples do not depend on the launch schedule.  n_levels = 0: plain tracing.  Each level is one launch of
 * the looping kernel: 64 / samples rays per 64-row tile, a pool of sdfr_trace_pool() persistent workgroups fetching tiles from a device counter,
 * survivors appended to the next level's list -- fewer, equally long passes for the grazing rays that end a march.
 * Decoders with LayerNorm or a hidden width below 257 march with per-step launches of their own float32 forward kernels (plain tracing, no
 * looping kernel; half must be 0).
 * pix0/lam0 (filled by sdfr_trace_setup) and pix1/lam1 are the ping-pong active lists (lam: ray state float[n][4]), pix2/lam2 the hand-over
 * list between levels (same sizes; may be NULL with fewer than two levels; pix0/lam0 is the other one), sdf float[B*W*H] scratch, tail_rows_buf
 * float[sdfr_trace_pool()][64][L + 3] scratch of the looping kernel. */
int sdfr_trace_pool(void);
int sdfr_trace_march(const sdfr_decoder* dec, const float* pose, const float* Kinv, const float* latn, int L, int B, int W, int H, float eps,
                     int steps, int head_steps, int tail_rows, const int32_t* levels, int n_levels, float q_max, float sigma, int half,
                     int32_t* counters, int32_t* pix0, float* lam0, int32_t* pix1, float* lam1, int32_t* pix2, float* lam2, const float* far,
                     float* inputs, float* sdf, float* tail_rows_buf, float* hit_lam, float* hit_sdf, void* stream);
/* hit pixels (hit_lam > 0) -> compact list: rows float[n][L+3] = [latn, o + lam d] for sdfr_mlp_jacobian (rows_per_crop = B*W*H, B = 1,
 * idx = the identity written here, cnt = n_hits), hit_slot int32[B*W*H] = list position of the pixel's hit or -1.  n_hits: device int32,
 * zero on entry (counters + 6 after sdfr_trace_setup). */
int sdfr_trace_hits(const float* pose, const float* Kinv, const float* latn, int L, int B, int W, int H, const float* hit_lam, int32_t* n_hits,
                    int32_t* hit_slot, int32_t* idx, float* rows, void* stream);
/* images of the hits after one Newton step along the ray with the exact-f32 decoder value f0 [slot] and input Jacobian J [slot][L+3]
 * (non-grazing rays only): depth [B][1][H][W] = lam_s r_z, color [B][3][H][W] = NOCS of the hit point, normals [B][3][H][W] = (R n + 1) / 2,
 * mask [B][1][H][W] = 1; zero at the other pixels.  lam_s (optional) float[B*W*H]: the polished ray parameter. */
int sdfr_trace_composite(const float* pose, const float* Kinv, int L, int B, int W, int H, const float* hit_lam, const int32_t* hit_slot,
                         const float* J, const float* f0, float* color, float* mask, float* depth, float* normals, float* lam_s, void* stream);
/* backward at a fixed hit set through the implicit function f(o + lam d, z) = 0: image gradients (each may be NULL) -> g_pose [B][16]
 * (gradient w.r.t. the entries of the row-major 4x4 pose) and g_latn [B][L] (w.r.t. the normalised latent); fixed-order sums per crop.
 * ws: sdfr_trace_backward_ws_floats(B, W, H) floats.  sdfr_params_backward maps the result to yaw / trans / latent. */
int64_t sdfr_trace_backward_ws_floats(int B, int W, int H);
int sdfr_trace_backward(const float* pose, const float* Kinv, int L, int B, int W, int H, const float* hit_lam, const int32_t* hit_slot,
                        const float* J, const float* f0, const float* g_color, const float* g_depth, const float* g_normals, float* ws,
                        float* g_pose, float* g_latn, void* stream);

/* Ragged extents for the sphere tracer (r04): every crop of the batch its own image size and intrinsics, as for the splat path's `_r` entry points.
 * ext->wh: DEVICE int32[B][2] = (W_b, H_b); ext->pix_stride: pixels of a crop's slot in every per-pixel array (far, hit_lam, hit_sdf, hit_slot,
 * pt_slot, lam_s: [B * pix_stride]; images [B][C][pix_stride]; global pixel id = b * pix_stride + y * W_b + x); ext->cone_cap: cone slots per crop
 * (>= ceil(W_b / block) * ceil(H_b / block); cone float[B * cone_cap], the cone lists n = B * cone_cap).  The struct itself is read on the HOST at
 * the call.  Same arithmetic per crop as the fixed-extent entry points: a crop traces bit-identically alone at its own size. */
typedef struct sdfr_extents { const int32_t* wh; int pix_stride; int cone_cap; } sdfr_extents;
int sdfr_trace_setup_r(const float* pose, const float* Kinv, const float* latn, int L, int B, const sdfr_extents* ext, float bound, float near,
                       int32_t* counters, int32_t* pix, float* lam, float* far, float* inputs, const float* cone, int cone_block, float* hit_lam,
                       float* hit_sdf, void* stream);
int sdfr_trace_cone_r(const sdfr_decoder* dec, const float* pose, const float* Kinv, const float* latn, int L, int B, const sdfr_extents* ext,
                      float bound, float near, float eps, int block, int cone_steps, int spec_k, float sigma, int half, int32_t* counters,
                      int32_t* ids0, float* st0, float* aux0, int32_t* ids1, float* st1, float* aux1, float* inputs, float* sdf, float* cone,
                      void* stream);
int sdfr_trace_march_r(const sdfr_decoder* dec, const float* pose, const float* Kinv, const float* latn, int L, int B, const sdfr_extents* ext, float eps,
                       int steps, int head_steps, int tail_rows, const int32_t* levels, int n_levels, float q_max, float sigma, int half,
                       int32_t* counters, int32_t* pix0, float* lam0, int32_t* pix1, float* lam1, int32_t* pix2, float* lam2, const float* far,
                       float* inputs, float* sdf, float* tail_rows_buf, float* hit_lam, float* hit_sdf, void* stream);
int sdfr_trace_hits_r(const float* pose, const float* Kinv, const float* latn, int L, int B, const sdfr_extents* ext, const float* hit_lam,
                      int32_t* n_hits, int32_t* hit_slot, int32_t* idx, float* rows, void* stream);
int sdfr_trace_composite_r(const float* pose, const float* Kinv, int L, int B, const sdfr_extents* ext, const float* hit_lam, const int32_t* hit_slot,
                           const float* J, const float* f0, float* color, float* mask, float* depth, float* normals, float* lam_s, void* stream);
int sdfr_trace_points_r(const float* Kinv, int B, const sdfr_extents* ext, const int32_t* hit_slot, const float* lam_s, float* xyzf, int ecap,
                        int32_t* ecnt, int32_t* pt_slot, void* stream);
int sdfr_trace_refine_backward_r(const float* pose, const float* Kinv, int L, int B, const sdfr_extents* ext, const float* hit_lam,
                                 const int32_t* hit_slot, const float* J, const float* f0, const float* g_color, const float* g_depth,
                                 const float* g_normals, const float* g_xyzf, const int32_t* pt_slot, int ecap, int surfel, float* ws /* sdfr_trace_backward_ws_floats(B, pix_stride, 1) */,
                                 float* g_pose, float* g_latn, void* stream);

/* The sphere tracer as a backend of the refinement loop (pipelines/optimizer.py:110-146 reads rendering['color'] and points['xyzf'] from its
 * renderer).  sdfr_trace_points gives points['xyzf'] of a traced render: the camera-frame hit points p_cam = lam_s K^-1 [x, y, 1] of each crop's
 * hit pixels, compacted in pixel (row-major) order into xyzf [B][ecap][3]; ecnt [B] = the crop's TRUE hit count (surplus beyond ecap dropped:
 * callers compare), pt_slot int32[B*W*H] = a pixel's row or -1.  hit_slot / lam_s: from sdfr_trace_hits / sdfr_trace_composite. */
int sdfr_trace_points(const float* Kinv, int B, int W, int H, const int32_t* hit_slot, const float* lam_s, float* xyzf, int ecap,
                      int32_t* ecnt, int32_t* pt_slot, void* stream);
/* sdfr_trace_backward with the gradient arriving through those points (g_xyzf [B][ecap][3] + pt_slot, both may be NULL) and a choice of
 * derivative semantics:
 *   surfel = 0  image-space derivative at the fixed pixels through the implicit function f(o + lam d, z) = 0 (what sdfr_trace_backward computes;
 *               a hit point then moves along its fixed pixel ray only: g_lam += g_xyzf . r);
 *   surfel = 1  the autograd semantics of the reference's surface points (sdfrenderer/grid.py:61 p = x - sdf n_hat with n_hat constant;
 *               sdfrenderer/renderer/projection.py:53-58 colour = the point's own object coordinates, p_cam = R p + t): every hit is a MATERIAL
 *               point x_s that moves rigidly with the pose and along its normal with the latent, d x_s = -n_hat (d f/d z . dz) / |grad f|; its
 *               NOCS colour does not depend on the pose; depth = (R x_s + t)_z; normals = (R n_hat + 1) / 2 with n_hat constant.
 *               This is the mode the refinement loop converges with (the loop's one-directional nearest-neighbour 3-D loss needs points that
 *               can move laterally). */
int sdfr_trace_refine_backward(const float* pose, const float* Kinv, int L, int B, int W, int H, const float* hit_lam, const int32_t* hit_slot,
                               const float* J, const float* f0, const float* g_color, const float* g_depth, const float* g_normals,
                               const float* g_xyzf, const int32_t* pt_slot, int ecap, int surfel, float* ws, float* g_pose, float* g_latn,
                               void* stream);

/* The decoder's scale head on one latent row (deep_sdf_decoder_scale.py:68-75,110-112): out[0] = W3 relu(W2 relu(W1 lat + b1) + b2) + b3 with
 * W1 [3][L], W2 [3][3], W3 [1][3] row-major as nn.Linear stores them.  Returned by Decoder.forward next to the SDF values; unused by the loop. */
int sdfr_scale_net(const float* latent_row, int L, const float* W1, const float* b1, const float* W2, const float* b2, const float* W3,
                   const float* b3, float* out, void* stream);

/* Debug only: forward kernels of a library built with -DSDFR_MLP_TRACE write cycle stamps of their workgroup 0 into this device buffer
 * (2 * SDFR_MAX_LAYERS * 5 uint64; see tools/cycle_trace.py); pass NULL to disable.  Production builds ignore it. */
int sdfr_debug_set_trace(void* device_buffer);

#ifdef __cplusplus
}
#endif
#endif /* SDFR_H */
