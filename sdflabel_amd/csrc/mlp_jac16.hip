// Band Jacobian of the float16 decoder, padded hidden width 512: mask-fed backward (MODE 3) with half operands on
// v_mfma_f32_16x16x32_f16 (float32 accumulation), 16-point workgroups.  The in-gradients are rounded to half between layers like the
// forward's activations; the transposed half weight image is half the bytes of the float32 one, and the kernel is paced by that stream.
// Geometry macros (tools/ab_variant.sh): SDFR_J16_FT feature tiles per wave, SDFR_J16_NW waves (16 * FT * NW = 512), SDFR_J16_PF ring.
#include "mlp_kernel.h"
#include <stdlib.h>
#include <string.h>
#ifndef SDFR_J16_PF
#define SDFR_J16_PF 4
#endif
#ifndef SDFR_J16_FT
#define SDFR_J16_FT 4
#define SDFR_J16_NW 8
#endif
#ifndef SDFR_J16_SWITCH_CROPS
#define SDFR_J16_SWITCH_CROPS 2       // 64-row tiles from this many crops per launch (tools/jac16_time.py, us per launch, 16-row / 64-row tiles:
#endif                                // 1 crop 50 / 81, 2 crops 97 / 80, 4: 144 / 81, 8: 287 / 157, 64: 2148 / 1090)
void sdfr_launch_jac_f16_512_many(const MlpParams& P, int cap, int B, hipStream_t s);
// (r06: 32x32x16 products on 32- / 64- / 128-row tiles were measured for this kernel and rejected -- 1.2 / 1.1 / 2.1 ms at 64 crops against 0.98;
// profiles/r06_notes.md section 2.  The hook that selected them is gone: they summed in another order, i.e. gave other bits.)
static int j16_pool_crops() {            // (SDFR_J16_POOL_CROPS: A/B hook; a huge value turns the pool off)
    static const int v = [] { const char* e = getenv("SDFR_J16_POOL_CROPS"); return e ? atoi(e) : 12; }();
    return v;
}
void sdfr_launch_jac_f16_512(const MlpParams& P, int cap, int B, hipStream_t s) {
    static_assert(16 * SDFR_J16_FT * SDFR_J16_NW == 512, "padded width 512 = 16 * FT * NW");
    if (B >= SDFR_J16_SWITCH_CROPS) { sdfr_launch_jac_f16_512_many(P, cap, B, s); return; }
    const dim3 grid(sdfr_cdiv(cap, 16), B);
    hipLaunchKernelGGL((sdfr_mlp_kernel<h16, 16, SDFR_J16_FT, 1, SDFR_J16_NW, SDFR_J16_PF, 3>), grid, dim3(64 * SDFR_J16_NW), 0, s, P);
}

// The same backward on 64-row tiles (v_mfma_f32_32x32x16_f16, two 32-point tiles per workgroup): launches with many thousands of selected
// rows (two or more crops' bands; the sphere tracer's hit pass: 18 k - 73 k hits), where 16-row tiles pay the weight stream of a tile per 16
// rows.  One 32-point tile per workgroup (-DSDFR_J16_MANY_NP=1) measured 59 / 61 / 118 / 177 / 1340 us at 1 / 2 / 4 / 8 / 64 crops.
#ifndef SDFR_J16_MANY_NP
#define SDFR_J16_MANY_NP 2
#endif
#ifndef SDFR_J16_MANY_PF
#define SDFR_J16_MANY_PF 2
#endif
void sdfr_launch_jac_f16_512_many(const MlpParams& P, int cap, int B, hipStream_t s) {
    const dim3 grid(sdfr_cdiv(cap, 32 * SDFR_J16_MANY_NP), B);
#ifdef SDFR_J16_MANY_32
    hipLaunchKernelGGL((sdfr_mlp_kernel<h16, 32, 2, SDFR_J16_MANY_NP, 8, 2, 3>), grid, dim3(512), 0, s, P);
#else
    // 16x16x32 products like the 16-row kernel -- every in-gradient is accumulated in the same order, so a crop's Jacobian has the same bits
    // whichever geometry the launch picked (batch-independent results) -- on 2 * NP point tiles of 16 per workgroup
    // r06: from 8 crops (and at most 64, the pool's prefix table) a POOL of one workgroup per CU walks the live band tiles back to back -- the
    // tiles take 130 KB of LDS, so a CU holds one workgroup, and as one-tile workgroups each of a CU's ~11 tiles paid its own dispatch
    // (from two full rounds of tile slots: at 8 crops -- 344 live tiles -- the pool measured 119 us against 111)
    const int64_t tiles = (int64_t)sdfr_cdiv(cap, 32 * SDFR_J16_MANY_NP) * B;
    if (B >= j16_pool_crops() && B <= 64 && P.n_crops == B) {
        hipLaunchKernelGGL((sdfr_mlp_kernel<h16, 16, SDFR_J16_FT, 2 * SDFR_J16_MANY_NP, SDFR_J16_NW, SDFR_J16_MANY_PF, 3, 0, false, 2>),
                           dim3((unsigned)(tiles < 256 ? tiles : 256)), dim3(64 * SDFR_J16_NW), 0, s, P);
        return;
    }
    hipLaunchKernelGGL((sdfr_mlp_kernel<h16, 16, SDFR_J16_FT, 2 * SDFR_J16_MANY_NP, SDFR_J16_NW, SDFR_J16_MANY_PF, 3>), grid, dim3(64 * SDFR_J16_NW), 0, s, P);
#endif
}

// Forward with half operands on 16-row tiles (MODE 0): the thin steps of the sphere tracer's march with the float16 decoder -- one
// decoder pass of latency per workgroup, paced by the 3.7 MB weight stream of a tile instead of the 128-point tile's matrix work.
void sdfr_launch_fwd_f16_512_tile16(const MlpParams& P, int64_t n, hipStream_t s) {
    hipLaunchKernelGGL((sdfr_mlp_kernel<h16, 16, SDFR_J16_FT, 1, SDFR_J16_NW, SDFR_J16_PF, 0>), dim3(sdfr_cdiv(n, 16)), dim3(64 * SDFR_J16_NW), 0, s,
                       P);
}

// MODE 4: the persistent tail of the sphere tracer's march with half operands (see mlp_jac.hip).  A pass of a tile is paced by the weight
// stream L2 -> CU (3.6 MB per pass), not by the matrix work, up to ~64 rows: spec_k = 4 (64 rows: 16 rays x 4 samples, the grid forward's 32x32
// tiles, two per workgroup) costs what spec_k = 1 (16 rows) costs per pass.
#ifndef SDFR_T16_FT
#define SDFR_T16_FT SDFR_J16_FT
#define SDFR_T16_NW SDFR_J16_NW
#define SDFR_T16_PF SDFR_J16_PF
#endif
#ifndef SDFR_T64_PF
#define SDFR_T64_PF 2          // weight-fragment ring of the 64-row half kernels (forward on 64-row tiles, looping tail): A/B with tools/ab_t64.sh
#define SDFR_T64_PFB 2
#endif
void sdfr_launch_tail_f16_512(const MlpParams& P, int64_t n_rays, int spec_k, hipStream_t s) {
    static_assert(16 * SDFR_T16_FT * SDFR_T16_NW == 512, "padded width 512 = 16 * FT * NW");
    const dim3 grid(sdfr_cdiv(n_rays, P.t_rt));           // a tile = t_rt rays (16: K <= 4 samples per pass; 8: K <= 8; 4: K <= 16) on 64 rows
    if (spec_k > 1)
        hipLaunchKernelGGL((sdfr_mlp_kernel<h16, 32, 2, 2, 8, SDFR_T64_PF, 4, SDFR_T64_PFB>), grid, dim3(512), 0, s, P);
    else
        hipLaunchKernelGGL((sdfr_mlp_kernel<h16, 16, SDFR_T16_FT, 1, SDFR_T16_NW, SDFR_T16_PF, 4>), grid, dim3(64 * SDFR_T16_NW), 0, s, P);
}

// Half forward on 64-row tiles (two 32-point tiles per workgroup instead of the grid forward's four): the middle steps of the sphere tracer's
// march, when the active rays fill the chip once with 64-row tiles but only a fraction of it with 128-row tiles -- a step is then one pass of
// a half-size tile (~60 us) instead of one pass of a full-size tile (~100 us).
void sdfr_launch_fwd_f16_512_tile64(const MlpParams& P, int64_t n, hipStream_t s) {
    hipLaunchKernelGGL((sdfr_mlp_kernel<h16, 32, 2, 2, 8, SDFR_T64_PF, 0, SDFR_T64_PFB>), dim3(sdfr_cdiv(n, 64)), dim3(512), 0, s, P);
}
