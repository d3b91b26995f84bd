// Decoder forward on the grid with float16 operands (f32 accumulate), padded hidden width 512, 128-point tiles; compiled alone.
// Geometry macros (tools/ab_build.sh A/B builds): SDFR_H_FT feature tiles per wave, SDFR_H_NW waves (FT*NW = 16), SDFR_H_PF / _PFB rings.
#include "mlp_kernel.h"
#ifndef SDFR_H_FT
#define SDFR_H_FT 2
#define SDFR_H_NW 8
#define SDFR_H_PF 2
#define SDFR_H_PFB 2
#endif
#ifndef SDFR_H_NP
#define SDFR_H_NP 4            // point tiles (32 points) per workgroup
#endif
int sdfr_fwd_f16_512_np() { return SDFR_H_NP; }
// ... on half-size tiles (64 points, masks in the layout of fwd_np = 2): launches of a few thousand rows -- the candidate rows of one or two crops --
// are a few dozen 128-row tiles on 256 CUs, and a tile pass is pure latency.  The same product shape and k order per point tile: every row gets
// the bits of the 128-row launch (tests/test_gpu_candidate_reuse.py).
void sdfr_launch_fwd_f16_512_half_tiles(const MlpParams& P, int64_t n, hipStream_t s) {
    hipLaunchKernelGGL((sdfr_mlp_kernel<h16, 32, SDFR_H_FT, 2, SDFR_H_NW, SDFR_H_PF, 1, SDFR_H_PFB>), dim3(sdfr_cdiv(n, 64)), dim3(64 * SDFR_H_NW), 0, s, P);
}
void sdfr_launch_fwd_f16_512(const MlpParams& P, int64_t n, bool save_masks, hipStream_t s) {
    const int grid = sdfr_cdiv(n, 32 * SDFR_H_NP);
    static_assert(SDFR_H_FT * SDFR_H_NW == 16, "padded width 512 = 32 * FT * NW");
    if (save_masks)
        hipLaunchKernelGGL((sdfr_mlp_kernel<h16, 32, SDFR_H_FT, SDFR_H_NP, SDFR_H_NW, SDFR_H_PF, 1, SDFR_H_PFB>), dim3(grid), dim3(64 * SDFR_H_NW), 0, s, P);
    else
        hipLaunchKernelGGL((sdfr_mlp_kernel<h16, 32, SDFR_H_FT, SDFR_H_NP, SDFR_H_NW, SDFR_H_PF, 0, SDFR_H_PFB>), dim3(grid), dim3(64 * SDFR_H_NW), 0, s, P);
}
