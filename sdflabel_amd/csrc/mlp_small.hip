// Decoders with padded hidden width 128 or 256 (small nets: tests, ablations): 4 waves, all modes.
#include "mlp_kernel.h"
void sdfr_launch_small(const MlpParams& P, int HP, int mode, int gx, int gy, hipStream_t s) {
    const dim3 g(gx, gy), b(256);
    if (HP == 128) {
        switch (mode) {
            case 0: hipLaunchKernelGGL((sdfr_mlp_kernel<float, 32, 1, 2, 4, 2, 0>), g, b, 0, s, P); break;
            case 1: hipLaunchKernelGGL((sdfr_mlp_kernel<float, 32, 1, 2, 4, 2, 1>), g, b, 0, s, P); break;
            case 2: hipLaunchKernelGGL((sdfr_mlp_kernel<float, 32, 1, 1, 4, 2, 2>), g, b, 0, s, P); break;
            default: hipLaunchKernelGGL((sdfr_mlp_kernel<float, 32, 1, 1, 4, 2, 3>), g, b, 0, s, P); break;
        }
    } else {
        switch (mode) {
            case 0: hipLaunchKernelGGL((sdfr_mlp_kernel<float, 32, 2, 2, 4, 2, 0>), g, b, 0, s, P); break;
            case 1: hipLaunchKernelGGL((sdfr_mlp_kernel<float, 32, 2, 2, 4, 2, 1>), g, b, 0, s, P); break;
            case 2: hipLaunchKernelGGL((sdfr_mlp_kernel<float, 32, 2, 1, 4, 2, 2>), g, b, 0, s, P); break;
            default: hipLaunchKernelGGL((sdfr_mlp_kernel<float, 32, 2, 1, 4, 2, 3>), g, b, 0, s, P); break;
        }
    }
}
