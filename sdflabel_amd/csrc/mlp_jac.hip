// Band Jacobian, float32, padded hidden width 512: 16x16x4 MFMA tiles, 16-point workgroups; mask-fed (MODE 3) or recomputing (MODE 2).
#include "mlp_kernel.h"
#ifndef SDFR_JAC_PF
#define SDFR_JAC_PF 4
#endif
void sdfr_launch_jac_f32_512(const MlpParams& P, int cap, int B, bool from_masks, hipStream_t s) {
    const dim3 grid(sdfr_cdiv(cap, 16), B);
    if (from_masks) hipLaunchKernelGGL((sdfr_mlp_kernel<float, 16, 4, 1, 8, SDFR_JAC_PF, 3>), grid, dim3(512), 0, s, P);
    else hipLaunchKernelGGL((sdfr_mlp_kernel<float, 16, 4, 1, 8, SDFR_JAC_PF, 2>), grid, dim3(512), 0, s, P);
}
