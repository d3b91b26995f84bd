// Decoder forward on the grid with float16 operands (f32 accumulate), padded hidden width 512, 128-point tiles; compiled alone.
#include "mlp_kernel.h"
void sdfr_launch_fwd_f16_512(const MlpParams& P, int grid, bool save_masks, hipStream_t s) {
    if (save_masks) hipLaunchKernelGGL((sdfr_mlp_kernel<h16, 32, 2, 4, 8, 2, 1>), dim3(grid), dim3(512), 0, s, P);
    else hipLaunchKernelGGL((sdfr_mlp_kernel<h16, 32, 2, 4, 8, 2, 0>), dim3(grid), dim3(512), 0, s, P);
}
