// Band Jacobian, float32, padded hidden width 512; mask-fed (MODE 3) or recomputing (MODE 2).
//   few crops per launch   16-row tiles (16x16x4 MFMA), 4 waves each owning 128 features (one wave per SIMD): a few thousand band rows of ONE
//                          crop give ~170 workgroups for 256 CUs; every workgroup streams the whole transposed weight image, so the kernel
//                          sits between its matrix floor (111 us) and its L2 stream (172 x 7.2 MB): 136 us (8 waves of 64 features: 158 us)
//   many crops per launch  32-row tiles (32x32x2 MFMA, paired K order = the 16-row kernel's, see gemm_pair), 4 waves of 128 features:
//                          half the weight stream per row, 118 TFLOP/s = 75 % of the f32 MFMA peak at 64 crops (16-row tiles: 93-108)
// The two return identical bits (same k order, same first-layer partition), so the choice is invisible in the results.
// Geometry macros (tools/ab_variant.sh A/B builds): SDFR_JAC_MS/_FT/_NP/_NW/_PF override the many-crops variant, SDFR_JAC_SMALL_* the other.
#include "mlp_kernel.h"
#ifndef SDFR_JAC_MS
#define SDFR_JAC_MS 32
#define SDFR_JAC_FT 4
#define SDFR_JAC_NP 1
#define SDFR_JAC_NW 4
#define SDFR_JAC_PF 2
#endif
#ifndef SDFR_JAC_SMALL_FT
#define SDFR_JAC_SMALL_FT 8
#define SDFR_JAC_SMALL_NW 4
#define SDFR_JAC_SMALL_PF 4
#endif
#ifndef SDFR_JAC_SWITCH_ROWS
#define SDFR_JAC_SWITCH_ROWS 8       // 32-row tiles from this many crops per launch (measured: 6 crops 665 vs 705 us, 8 crops 798 vs 712 us)
#endif
// recomputing Jacobian on 32-row tiles: half the weight stream per row of the 16-row variant -- for launches of many thousands of rows (the
// sphere tracer's hits), where tiles outnumber the CUs several times and the stream, not one tile's latency, sets the time
void sdfr_launch_jac_f32_512_recompute32(const MlpParams& P, int cap, int B, hipStream_t s) {
    hipLaunchKernelGGL((sdfr_mlp_kernel<float, 32, 4, 1, 4, 2, 2>), dim3(sdfr_cdiv(cap, 32), B), dim3(256), 0, s, P);
}

#include <stdlib.h>
static int jac_pool_crops() {            // (SDFR_JAC_POOL_CROPS: A/B hook; a huge value turns the pool off)
    static const int v = [] { const char* e = getenv("SDFR_JAC_POOL_CROPS"); return e ? atoi(e) : 12; }();
    return v;
}
void sdfr_launch_jac_f32_512(const MlpParams& P, int cap, int B, bool from_masks, hipStream_t s) {
    static_assert(SDFR_JAC_MS * SDFR_JAC_FT * SDFR_JAC_NW == 512 && 16 * SDFR_JAC_SMALL_FT * SDFR_JAC_SMALL_NW == 512, "padded width 512 = MS * FT * NW");
    if (!from_masks) {
        // recomputing Jacobian (no saved masks): 16-row tiles, 4 waves x 128 features (542 -> 511 us for 4371 rows against 8 waves x 64)
        hipLaunchKernelGGL((sdfr_mlp_kernel<float, 16, 8, 1, 4, 4, 2>), dim3(sdfr_cdiv(cap, 16), B), dim3(256), 0, s, P);
    } else if (B >= jac_pool_crops() && B <= 64 && P.n_crops == B) {
        // r06: a pool of two workgroups per CU (67 KB of LDS each) walks the crops' live band tiles back to back (mlp_kernel.h, JPOOL): no
        // dispatch between a CU's ~21 tiles, no dead tile slots (a launch is sized for the capacity: 60 % of its slots are beyond the counts)
        const int64_t tiles = (int64_t)sdfr_cdiv(cap, SDFR_JAC_MS * SDFR_JAC_NP) * B;
        hipLaunchKernelGGL((sdfr_mlp_kernel<float, SDFR_JAC_MS, SDFR_JAC_FT, SDFR_JAC_NP, SDFR_JAC_NW, SDFR_JAC_PF, 3, 0, false, 2>),
                           dim3((unsigned)(tiles < 512 ? tiles : 512)), dim3(64 * SDFR_JAC_NW), 0, s, P);
    } else if (B >= SDFR_JAC_SWITCH_ROWS) {
        hipLaunchKernelGGL((sdfr_mlp_kernel<float, SDFR_JAC_MS, SDFR_JAC_FT, SDFR_JAC_NP, SDFR_JAC_NW, SDFR_JAC_PF, 3>),
                           dim3(sdfr_cdiv(cap, SDFR_JAC_MS * SDFR_JAC_NP), B), dim3(64 * SDFR_JAC_NW), 0, s, P);
    } else {
        hipLaunchKernelGGL((sdfr_mlp_kernel<float, 16, SDFR_JAC_SMALL_FT, 1, SDFR_JAC_SMALL_NW, SDFR_JAC_SMALL_PF, 3>), dim3(sdfr_cdiv(cap, 16), B),
                           dim3(64 * SDFR_JAC_SMALL_NW), 0, s, P);
    }
}

// Forward on 16-row tiles (the geometry of the band kernels above, MODE 0): a launch of at most a few thousand rows -- the thin steps of the
// sphere tracer's march -- is one decoder pass of LATENCY per workgroup, and a 16-row tile's pass is 0.12 ms where a 64-row tile's is 0.44 ms.
void sdfr_launch_fwd_f32_512_tile16(const MlpParams& P, int64_t n, hipStream_t s) {
    hipLaunchKernelGGL((sdfr_mlp_kernel<float, 16, SDFR_JAC_SMALL_FT, 1, SDFR_JAC_SMALL_NW, SDFR_JAC_SMALL_PF, 0>), dim3(sdfr_cdiv(n, 16)),
                       dim3(64 * SDFR_JAC_SMALL_NW), 0, s, P);
}

// MODE 4: the persistent tail of the sphere tracer's march (csrc/trace.hip sdfr_trace_march): each workgroup marches the t_rt rays of its
// tile -- decoder pass, step rule, hit / exit test, next pass -- without leaving the kernel.  spec_k = 1: 16 rays on 16 operand rows per tile
// (the band kernels' geometry); spec_k > 1: 64 rows (4 point tiles of 16) = t_rt rays x up to 64 / t_rt samples per ray and pass.
void sdfr_launch_tail_f32_512(const MlpParams& P, int64_t n_rays, int spec_k, hipStream_t s) {
    const dim3 grid(sdfr_cdiv(n_rays, P.t_rt));
    if (spec_k > 1)
        hipLaunchKernelGGL((sdfr_mlp_kernel<float, 16, SDFR_JAC_SMALL_FT, 4, SDFR_JAC_SMALL_NW, 2, 4>), grid, dim3(64 * SDFR_JAC_SMALL_NW), 0, s, P);
    else
        hipLaunchKernelGGL((sdfr_mlp_kernel<float, 16, SDFR_JAC_SMALL_FT, 1, SDFR_JAC_SMALL_NW, SDFR_JAC_SMALL_PF, 4>), grid, dim3(64 * SDFR_JAC_SMALL_NW), 0,
                           s, P);
}
