// Decoder forward on the grid, float32, padded hidden width 512 -- the dominant kernel of the whole path, compiled alone.
// 8 waves, 64-point tiles; weight-fragment ring SDFR_FWD_PF, activation-fragment ring SDFR_FWD_PFB (tools/ab_build.py A/B builds).
#include "mlp_kernel.h"
#ifndef SDFR_FWD_PF
#define SDFR_FWD_PF 8
#endif
#ifndef SDFR_FWD_PFB
#define SDFR_FWD_PFB 2
#endif
void sdfr_launch_fwd_f32_512(const MlpParams& P, int grid, bool save_masks, hipStream_t s) {
    if (save_masks) hipLaunchKernelGGL((sdfr_mlp_kernel<float, 32, 2, 2, 8, SDFR_FWD_PF, 1, SDFR_FWD_PFB>), dim3(grid), dim3(512), 0, s, P);
    else hipLaunchKernelGGL((sdfr_mlp_kernel<float, 32, 2, 2, 8, SDFR_FWD_PF, 0, SDFR_FWD_PFB>), dim3(grid), dim3(512), 0, s, P);
}
