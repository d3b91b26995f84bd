// Decoder forward, padded hidden width 512, as a POOL of workgroups that walk the live tiles of a launch (r06) -- the launches of the
// candidate-reuse modes, whose work is decided on the device:
//   * the candidates' pass: a ragged [B][stride] launch, crop b evaluating its first ccnt[b] rows.  GATHER: the rows are read where they lie
//     in the [B][G] input array through the candidate index list (no gathered copy, one launch less per iteration); PERSIST: the pool walks
//     the live tiles of all crops back to back -- dead tile slots (60 % of a launch sized for the capacity) are never dispatched;
//   * the full-grid pass with per-crop skip flags: on 58 of 60 iterations of a refinement every crop is flagged and the launch has nothing to
//     do; as a one-workgroup-per-tile launch that cost 13 us at one crop (500 workgroups of 140 KB LDS dispatched to exit), as a pool it is
//     one wave of dispatch.
// The same template, product shape, k order and per-tile arithmetic as the grid kernels (mlp_fwd16.hip / mlp_fwd32.hip): a row's value and
// masks have the bits of any other launch that evaluates it; these instantiations live in their own translation unit so that the hot grid
// kernels are compiled exactly as before.
#include "mlp_kernel.h"

#define SDFR_POOL_WGS 256          // one workgroup per CU (the operand tiles take 128-140 KB of LDS: one resident workgroup per CU)

static inline int pool_grid(int64_t tiles) { return (int)(tiles < SDFR_POOL_WGS ? tiles : SDFR_POOL_WGS); }

// float16, 128-row tiles (the grid forward's geometry <h16, 32, 2, 4, 8, 2, *, 2>)
void sdfr_launch_pool_f16_skip(const MlpParams& P, int64_t n, hipStream_t s) {
    hipLaunchKernelGGL((sdfr_mlp_kernel<h16, 32, 2, 4, 8, 2, 0, 2, false, 2>), dim3(pool_grid(sdfr_cdiv(n, 128))), dim3(512), 0, s, P);
}
void sdfr_launch_pool_f16_ragged(const MlpParams& P, int64_t n, bool gather, hipStream_t s) {
    if (gather) hipLaunchKernelGGL((sdfr_mlp_kernel<h16, 32, 2, 4, 8, 2, 1, 2, false, 3>), dim3(pool_grid(sdfr_cdiv(n, 128))), dim3(512), 0, s, P);
    else hipLaunchKernelGGL((sdfr_mlp_kernel<h16, 32, 2, 4, 8, 2, 1, 2, false, 2>), dim3(pool_grid(sdfr_cdiv(n, 128))), dim3(512), 0, s, P);
}
// float16, 64-row tiles (one or two crops per launch: twice the workgroups for the same rows, same bits per row)
void sdfr_launch_pool_f16_ragged_half_tiles(const MlpParams& P, int64_t n, bool gather, hipStream_t s) {
    if (gather) hipLaunchKernelGGL((sdfr_mlp_kernel<h16, 32, 2, 2, 8, 2, 1, 2, false, 3>), dim3(pool_grid(sdfr_cdiv(n, 64))), dim3(512), 0, s, P);
    else hipLaunchKernelGGL((sdfr_mlp_kernel<h16, 32, 2, 2, 8, 2, 1, 2, false, 2>), dim3(pool_grid(sdfr_cdiv(n, 64))), dim3(512), 0, s, P);
}
// float16, 32-row tiles (ONE crop per launch: ~100 tiles for its ~3 000 candidate rows instead of ~50 -- a tile pass is paced by the 3.6 MB weight
// stream of its CU, not by its rows, so halving the rows per tile shortens the pass and doubles the CUs at work; same bits per row)
void sdfr_launch_pool_f16_ragged_quarter_tiles(const MlpParams& P, int64_t n, hipStream_t s) {
    hipLaunchKernelGGL((sdfr_mlp_kernel<h16, 32, 2, 1, 8, 2, 1, 2, false, 3>), dim3(pool_grid(sdfr_cdiv(n, 32))), dim3(512), 0, s, P);
}
// exact float32, 64-row tiles (the grid forward's geometry <float, 32, 2, 2, 8, 2, 1, 2>)
void sdfr_launch_pool_f32_skip(const MlpParams& P, int64_t n, hipStream_t s) {
    hipLaunchKernelGGL((sdfr_mlp_kernel<float, 32, 2, 2, 8, 2, 1, 2, false, 2>), dim3(pool_grid(sdfr_cdiv(n, 64))), dim3(512), 0, s, P);
}
void sdfr_launch_pool_f32_ragged(const MlpParams& P, int64_t n, bool gather, hipStream_t s) {
    if (gather) hipLaunchKernelGGL((sdfr_mlp_kernel<float, 32, 2, 2, 8, 2, 1, 2, false, 3>), dim3(pool_grid(sdfr_cdiv(n, 64))), dim3(512), 0, s, P);
    else hipLaunchKernelGGL((sdfr_mlp_kernel<float, 32, 2, 2, 8, 2, 1, 2, false, 2>), dim3(pool_grid(sdfr_cdiv(n, 64))), dim3(512), 0, s, P);
}
