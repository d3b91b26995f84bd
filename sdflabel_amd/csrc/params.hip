// Parameter glue of the batched refinement step: optimizer parameters <-> renderer inputs, entirely on the device.
//
// Restates, for B crops at once, the host-side tensor algebra the reference's optimizer executes per iteration with dozens of
// tiny ATen launches (pipelines/optimizer.py:86-100): pose = [R_y(yaw) | t] with row 1 of the rotation negated (:87-90,
// utils/refinement.py:108-125), latent_ = F.normalize(latent, p=2, dim=0) (:96), inputs = cat(latent_.expand(G,-1), grid.points)
// (:99-100); and the matching backward.  Compiled with -ffp-contract=off.
#include "sdfr_common.h"

__global__ __launch_bounds__(256) void sdfr_params_forward_kernel(const float* __restrict__ yaw, const float* __restrict__ trans,
                                                                 const float* __restrict__ latent, int L,
                                                                 const float* __restrict__ grid, int64_t G,
                                                                 float* __restrict__ inputs, float* __restrict__ pose,
                                                                 float* __restrict__ latnorm) {
    const int b = blockIdx.y;
    const int NI = L + 3;
    // ||latent||_2, max(., 1e-12) as F.normalize does
    float ss = 0.f;
    for (int c = 0; c < L; ++c) ss += latent[b * L + c] * latent[b * L + c];
    const float nrm = fmaxf(sqrtf(ss), 1e-12f);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        latnorm[b] = nrm;
        const float c = cosf(yaw[b]), s = sinf(yaw[b]);
        float* P = pose + (int64_t)b * 16;
        P[0] = c;   P[1] = 0.f;  P[2] = s;   P[3] = trans[b * 3 + 0];
        P[4] = -0.f; P[5] = -1.f; P[6] = -0.f; P[7] = trans[b * 3 + 1];
        P[8] = -s;  P[9] = 0.f;  P[10] = c;  P[11] = trans[b * 3 + 2];
        P[12] = 0.f; P[13] = 0.f; P[14] = 0.f; P[15] = 1.f;
    }
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= G) return;
    float* row = inputs + ((int64_t)b * G + g) * NI;
    for (int c = 0; c < L; ++c) row[c] = latent[b * L + c] / nrm;
    row[L] = grid[g * 3]; row[L + 1] = grid[g * 3 + 1]; row[L + 2] = grid[g * 3 + 2];
}

extern "C" int sdfr_params_forward(const float* yaw, const float* trans, const float* latent, int L, const float* grid, int64_t G,
                                   int B, float* inputs, float* pose, float* latnorm, void* stream) {
    SDFR_REQUIRE(yaw && trans && latent && grid && inputs && pose && latnorm, "sdfr_params_forward: NULL argument");
    SDFR_REQUIRE(L >= 0 && L <= 1024 && G > 0, "sdfr_params_forward: bad size");
    if (B <= 0) return SDFR_OK;
    hipLaunchKernelGGL(sdfr_params_forward_kernel, dim3(sdfr_cdiv(G, 256), B), dim3(256), 0, (hipStream_t)stream, yaw, trans, latent,
                       L, grid, G, inputs, pose, latnorm);
    SDFR_LAUNCH_CHECK();
    return SDFR_OK;
}

// g_latn[b][c] = sum_s g_sdf_s * J[b][s][c],  g_sdf_s = -(g_points_s . n_hat_s)      (grid.py:61 backward, then the decoder's
// input gradient summed over the expanded latent rows).  One workgroup per crop, fixed-order tree (deterministic).
#define LAT_THREADS 1024
#define LAT_MAXL 8
__global__ __launch_bounds__(LAT_THREADS) void sdfr_surface_latent_grad_kernel(const float* __restrict__ g_points,
                                                                              const float* __restrict__ g_nocs,
                                                                              const float* __restrict__ normals,
                                                                              const float* __restrict__ J, int NI, int L, int cap,
                                                                              const int32_t* __restrict__ cnt,
                                                                              float* __restrict__ g_latn) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const int count = sdfr_count(cnt, b, cap);
    __shared__ float red[LAT_THREADS / 64];
    for (int c0 = 0; c0 < L; c0 += LAT_MAXL) {
        float acc[LAT_MAXL];
#pragma unroll
        for (int i = 0; i < LAT_MAXL; ++i) acc[i] = 0.f;
        for (int s = tid; s < count; s += LAT_THREADS) {
            const int64_t e = (int64_t)b * cap + s;
            float gx = g_points[e * 3], gy = g_points[e * 3 + 1], gz = g_points[e * 3 + 2];
            if (g_nocs) { gx += g_nocs[e * 3] / 2.f; gy += g_nocs[e * 3 + 1] / 2.f; gz += g_nocs[e * 3 + 2] / 2.f; }
            const float gs = -(gx * normals[e * 3] + gy * normals[e * 3 + 1] + gz * normals[e * 3 + 2]);
#pragma unroll
            for (int i = 0; i < LAT_MAXL; ++i)
                if (c0 + i < L) acc[i] += gs * J[e * NI + c0 + i];
        }
#pragma unroll
        for (int i = 0; i < LAT_MAXL; ++i) {
            if (c0 + i >= L) break;
            float v = acc[i];
            for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
            __syncthreads();
            if ((tid & 63) == 0) red[tid >> 6] = v;
            __syncthreads();
            if (tid == 0) {
                float t = 0.f;
                for (int w = 0; w < LAT_THREADS / 64; ++w) t += red[w];
                g_latn[b * L + c0 + i] = t;
            }
        }
    }
}

extern "C" int sdfr_surface_latent_grad(const float* g_points, const float* g_nocs, const float* normals, const float* J, int n_inputs,
                                        int L, int B, int cap, const int32_t* cnt, float* g_latn, void* stream) {
    SDFR_REQUIRE(g_points && normals && J && g_latn, "sdfr_surface_latent_grad: NULL argument");
    SDFR_REQUIRE(L >= 0 && L <= n_inputs, "sdfr_surface_latent_grad: bad latent size");
    if (B <= 0 || L == 0) return SDFR_OK;
    hipLaunchKernelGGL(sdfr_surface_latent_grad_kernel, dim3(B), dim3(LAT_THREADS), 0, (hipStream_t)stream, g_points, g_nocs, normals,
                       J, n_inputs, L, cap, cnt, g_latn);
    SDFR_LAUNCH_CHECK();
    return SDFR_OK;
}

// g_yaw, g_trans from g_pose (optimizer.py:87-90); g_latent from g_latn through F.normalize (optimizer.py:96)
__global__ __launch_bounds__(64) void sdfr_params_backward_kernel(const float* __restrict__ yaw, const float* __restrict__ latent, int L,
                                                                 const float* __restrict__ latnorm, const float* __restrict__ g_pose,
                                                                 const float* __restrict__ g_latn, int B, float* __restrict__ g_yaw,
                                                                 float* __restrict__ g_trans, float* __restrict__ g_latent) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    const float* g = g_pose + (int64_t)b * 16;
    const float c = cosf(yaw[b]), s = sinf(yaw[b]);
    // R = [[c,0,s],[0,-1,0],[-s,0,c]]  ->  dR/dyaw = [[-s,0,c],[0,0,0],[-c,0,-s]]
    g_yaw[b] = (-s) * g[0] + c * g[2] + (-c) * g[8] + (-s) * g[10];
    g_trans[b * 3] = g[3]; g_trans[b * 3 + 1] = g[7]; g_trans[b * 3 + 2] = g[11];
    const float nrm = latnorm[b];
    float dot = 0.f;
    for (int i = 0; i < L; ++i) dot += (latent[b * L + i] / nrm) * g_latn[b * L + i];
    for (int i = 0; i < L; ++i) g_latent[b * L + i] = (g_latn[b * L + i] - (latent[b * L + i] / nrm) * dot) / nrm;
}

extern "C" int sdfr_params_backward(const float* yaw, const float* latent, int L, const float* latnorm, const float* g_pose,
                                    const float* g_latn, int B, float* g_yaw, float* g_trans, float* g_latent, void* stream) {
    SDFR_REQUIRE(yaw && latent && latnorm && g_pose && g_latn && g_yaw && g_trans && g_latent, "sdfr_params_backward: NULL argument");
    if (B <= 0) return SDFR_OK;
    hipLaunchKernelGGL(sdfr_params_backward_kernel, dim3(sdfr_cdiv(B, 64)), dim3(64), 0, (hipStream_t)stream, yaw, latent, L, latnorm,
                       g_pose, g_latn, B, g_yaw, g_trans, g_latent);
    SDFR_LAUNCH_CHECK();
    return SDFR_OK;
}

// dst[b][idx[b][j]][:] += src[b][j][:] for j < cnt[b]  (gradient of the front-facing selection points_3d_filt, projection.py:64-70)
__global__ __launch_bounds__(256) void sdfr_scatter_add_rows3_kernel(float* __restrict__ dst, const float* __restrict__ src,
                                                                    const int32_t* __restrict__ idx, int cap,
                                                                    const int32_t* __restrict__ cnt) {
    const int b = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= sdfr_count(cnt, b, cap)) return;
    const int64_t d = ((int64_t)b * cap + idx[(int64_t)b * cap + j]) * 3, s = ((int64_t)b * cap + j) * 3;
    dst[d] += src[s]; dst[d + 1] += src[s + 1]; dst[d + 2] += src[s + 2];
}

extern "C" int sdfr_scatter_add_rows3(float* dst, const float* src, const int32_t* idx, int B, int cap, const int32_t* cnt,
                                      void* stream) {
    SDFR_REQUIRE(dst && src && idx, "sdfr_scatter_add_rows3: NULL argument");
    if (B <= 0 || cap <= 0) return SDFR_OK;
    hipLaunchKernelGGL(sdfr_scatter_add_rows3_kernel, dim3(sdfr_cdiv(cap, 256), B), dim3(256), 0, (hipStream_t)stream, dst, src, idx,
                       cap, cnt);
    SDFR_LAUNCH_CHECK();
    return SDFR_OK;
}

// out[b][j][:] = src[b][idx[b][j]][:] for j < cnt[b], zero beyond (points_3d_filt as a padded array)
__global__ __launch_bounds__(256) void sdfr_gather_rows3_kernel(float* __restrict__ out, const float* __restrict__ src,
                                                               const int32_t* __restrict__ idx, int cap,
                                                               const int32_t* __restrict__ cnt) {
    const int b = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= cap) return;
    const int64_t o = ((int64_t)b * cap + j) * 3;
    if (j < sdfr_count(cnt, b, cap)) {
        const int64_t s = ((int64_t)b * cap + idx[(int64_t)b * cap + j]) * 3;
        out[o] = src[s]; out[o + 1] = src[s + 1]; out[o + 2] = src[s + 2];
    } else {
        out[o] = 0.f; out[o + 1] = 0.f; out[o + 2] = 0.f;
    }
}

extern "C" int sdfr_gather_rows3(float* out, const float* src, const int32_t* idx, int B, int cap, const int32_t* cnt, void* stream) {
    SDFR_REQUIRE(out && src && idx, "sdfr_gather_rows3: NULL argument");
    if (B <= 0 || cap <= 0) return SDFR_OK;
    hipLaunchKernelGGL(sdfr_gather_rows3_kernel, dim3(sdfr_cdiv(cap, 256), B), dim3(256), 0, (hipStream_t)stream, out, src, idx, cap, cnt);
    SDFR_LAUNCH_CHECK();
    return SDFR_OK;
}
