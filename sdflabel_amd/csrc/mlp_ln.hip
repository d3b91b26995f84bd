// Decoders with LayerNorm after the hidden linears (weight_norm=False + norm_layers, reference deep_sdf_decoder_scale.py:56-57,99-101):
// float32 forward (MODE 0) and recomputing Jacobian (MODE 2, normalised pre-activations spilled to a global scratch).  A rarely used
// variant (DeepSDF checkpoints use weight norm): built for correctness, not tuned.
#include "mlp_kernel.h"
int sdfr_ln_points_per_wg(int HP, bool jac) { return jac ? (HP == 512 ? 16 : 32) : 64; }
void sdfr_launch_ln(const MlpParams& P, int HP, bool jac, int gx, int gy, hipStream_t s) {
    const dim3 g(gx, gy);
    if (HP == 128) {
        if (jac) hipLaunchKernelGGL((sdfr_mlp_kernel<float, 32, 1, 1, 4, 2, 2, 0, true>), g, dim3(256), 0, s, P);
        else hipLaunchKernelGGL((sdfr_mlp_kernel<float, 32, 1, 2, 4, 2, 0, 0, true>), g, dim3(256), 0, s, P);
    } else if (HP == 256) {
        if (jac) hipLaunchKernelGGL((sdfr_mlp_kernel<float, 32, 2, 1, 4, 2, 2, 0, true>), g, dim3(256), 0, s, P);
        else hipLaunchKernelGGL((sdfr_mlp_kernel<float, 32, 2, 2, 4, 2, 0, 0, true>), g, dim3(256), 0, s, P);
    } else {
        if (jac) hipLaunchKernelGGL((sdfr_mlp_kernel<float, 16, 4, 1, 8, 4, 2, 0, true>), g, dim3(512), 0, s, P);
        else hipLaunchKernelGGL((sdfr_mlp_kernel<float, 32, 2, 2, 8, 4, 0, 0, true>), g, dim3(512), 0, s, P);
    }
}
