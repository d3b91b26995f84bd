// Decoder forward on the grid, float32, padded hidden width 512 -- the dominant kernel of the whole path, compiled alone.
// Geometry macros (tools/ab_build.sh A/B builds): SDFR_FWD_FT feature tiles per wave, SDFR_FWD_NW waves (FT*NW = 16), SDFR_FWD_NP point
// tiles (32 points each) per workgroup, weight-fragment ring SDFR_FWD_PF, activation-fragment ring SDFR_FWD_PFB.
#include "mlp_kernel.h"
#ifndef SDFR_FWD_PF
#define SDFR_FWD_PF 2
#endif
#ifndef SDFR_FWD_PFB
#define SDFR_FWD_PFB 2
#endif
#ifndef SDFR_FWD_FT
#define SDFR_FWD_FT 2
#define SDFR_FWD_NW 8
#define SDFR_FWD_NP 2
#endif
int sdfr_fwd_f32_512_np() { return SDFR_FWD_NP; }
void sdfr_launch_fwd_f32_512(const MlpParams& P, int64_t n, bool save_masks, hipStream_t s) {
    static_assert(SDFR_FWD_FT * SDFR_FWD_NW == 16, "padded width 512 = 32 * FT * NW");
    const int grid = sdfr_cdiv(n, 32 * SDFR_FWD_NP);
    // one instantiation serves both cases: without a mask buffer the mask-saving kernel skips its stores (measured 1.78 ms against 1.93 ms
    // of a separate no-mask instantiation -- the compiler's schedule for that one is simply worse)
    (void)save_masks;
    hipLaunchKernelGGL((sdfr_mlp_kernel<float, 32, SDFR_FWD_FT, SDFR_FWD_NP, SDFR_FWD_NW, SDFR_FWD_PF, 1, SDFR_FWD_PFB>), dim3(grid),
                       dim3(64 * SDFR_FWD_NW), 0, s, P);
}
